"""Dev helper: free-running comparison -- reference and C restatement start from the same
state and are stepped independently; report first divergence."""
import sys, numpy as np, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from oracle.refblom import RefBackend
from oracle.coracle import COracle
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS
cfg = sys.argv[1]; nsteps = int(sys.argv[2])
c = make_case(cfg)
ref = RefBackend(cfg, c.depth)
hostinit.init_state(ref, c)
co = COracle(c.idm, c.jdm, c.kdm, ref.ntr, ref.nreg, ref.masks)
for nm, v in c.params.items():
    if not nm.endswith('0'): co.set(nm, v)
copy_state(ref, co)
co.set('delt1', c.params['baclin'])
ns_r = ns_c = 0
fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in ('util1','util2')]
t_ref = t_c = 0.
for it in range(nsteps):
    t0 = time.time(); ns_r = dyncore_step(ref, ns_r, c.params['baclin']); t_ref += time.time() - t0
    t0 = time.time(); ns_c = dyncore_step(co, ns_c, c.params['baclin']); t_c += time.time() - t0
    bad = diff_report(ref, co, fields=fields)
    if bad:
        print('step', ns_r, 'DIVERGED\n' + fmt_report(bad[:8])); break
else:
    print(f'{cfg}: {nsteps} free-running steps bit-identical; ref {t_ref/nsteps*1e3:.1f} ms/step, C {t_c/nsteps*1e3:.1f} ms/step')
