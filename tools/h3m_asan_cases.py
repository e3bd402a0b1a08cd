import sys, ctypes
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import h3m_cases as hc
hc.HOST_LIB = '/tmp/libh3m_hostcheck_asan.so'
import numpy as np
n = 0
for cfg in hc.CONFIGS:
    for ik, kind in enumerate(hc.KINDS):
        for rm in (hc.METHOD_1, hc.METHOD_2):
            for dec in (False, True):
                x, u, xd, ug = hc.make_columns(300 + ik, 120, 23, 19, 11, kind, dec)
                hc.run_hostcheck(*cfg, x, u, xd, ug, rm)
                n += 1
x, u, xd, ug = hc.make_slab(3, 500, 53, 53, 54)
for cfg in hc.CONFIGS:
    hc.run_hostcheck(*cfg, x, u, xd, ug, hc.METHOD_2)
print("asan run complete:", n, "cases")
