"""blom_amd -- MI355X-native dynamical core for the BLOM layered ocean model.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C-ABI library
libblomgpu.so), the ctypes host mirror of the reference's stage interface, and the
netCDF-free host initialisation / case generators.
"""
