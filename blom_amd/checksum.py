"""xccrc / chksum on the host (numpy + zlib): the reference's decomposition-independent field
checksum, phy/mod_xc.F90:4164-4205 with the CRC-32 of phy/mod_crc32.F90 (the zlib polynomial).
Used to compare device fields and fixtures with the `chksum:` values the reference prints under
csdiag (phy/mod_checksum.F90:41-74)."""
import zlib
import numpy as np

NBDY = 4
# grid of every field on the path: 1 p, 2 q, 3 u, 4 v (halo_ps/qs/us/vs, phy/mod_xc.F90:107-110)
U_FIELDS = {"u", "dpu", "uflx", "utflx", "usflx", "pu", "cau", "ub", "pbu", "ubflxs", "ubflxs_p", "pbu_p",
            "ubcors_p", "pgfx", "pgfx_o", "pgfxm", "xixp", "xixm", "pgfxm_o", "xixp_o", "xixm_o", "ubflx",
            "ubflx_mn", "dpuold", "umfltd", "umflsm", "utfltd", "utflsm", "utflld", "usfltd", "usflsm",
            "usflld", "utotm", "utotn", "umax", "taux"}
V_FIELDS = {"v", "dpv", "vflx", "vtflx", "vsflx", "pv", "cav", "vb", "pbv", "vbflxs", "vbflxs_p", "pbv_p",
            "vbcors_p", "pgfy", "pgfy_o", "pgfym", "xiyp", "xiym", "pgfym_o", "xiyp_o", "xiym_o", "vbflx",
            "vbflx_mn", "dpvold", "vmfltd", "vmflsm", "vtfltd", "vtflsm", "vtflld", "vsfltd", "vsflsm",
            "vsflld", "vtotm", "vtotn", "vmax", "tauy"}
Q_FIELDS = {"pvtrop"}


def grid_of(name):
    return 3 if name in U_FIELDS else 4 if name in V_FIELDS else 2 if name in Q_FIELDS else 1


def xccrc(a, mask, idm, jdm):
    """a: (nlev, nj, ni) float64; mask: (nj, ni) int.  Returns the unsigned CRC."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    o = NBDY - 1
    rows = np.zeros(jdm, dtype="<i4")
    at = np.ascontiguousarray(a.transpose(1, 2, 0))          # (nj, ni, nlev): a(i,j,:) contiguous
    for j in range(1, jdm + 1):
        crc8 = 0
        for i1 in range(1, idm + 1, 2 * NBDY + 1):
            crc8p = 0
            for i in range(i1, min(i1 + 2 * NBDY, idm) + 1):
                if mask[o + j, o + i] == 1:
                    crc8p = zlib.crc32(at[o + j, o + i].tobytes(), crc8p)
            crc8 = zlib.crc32(np.array([crc8p], dtype="<u4").tobytes(), crc8)
        rows[j - 1] = np.array([crc8], dtype="<u4").view("<i4")[0]
    return zlib.crc32(rows.tobytes()) & 0xFFFFFFFF


def chksum(name, a, masks, idm, jdm):
    g = grid_of(name)
    return xccrc(a, masks[{1: "ip", 2: "iq", 3: "iu", 4: "iv"}[g]], idm, jdm)
