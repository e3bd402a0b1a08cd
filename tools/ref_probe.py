"""Dev helper: run the reference dyncore sequence and report where NaN/Inf first appear."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, DYNCORE_STAGES
from oracle.refblom import RefBackend
cfg = sys.argv[1] if len(sys.argv)>1 else 'chan_s'
nsteps = int(sys.argv[2]) if len(sys.argv)>2 else 3
c = make_case(cfg)
be = RefBackend(cfg, c.depth)
hostinit.init_state(be, c)
names = ['u','v','dp','dpu','dpv','temp','saln','sigma','uflx','vflx','p','pu','pv','phi','pgfx','pgfy','pb','ub','vb','pbu','pbv','ubflxs','vbflxs','ubflxs_p','pb_p','pbu_p','ubcors_p','utotn','vtotn','cau','cav','trc']
def check(tag):
    bad = []
    for nm in names:
        a = be.get(nm)
        nb = int((~np.isfinite(a)).sum())
        if nb: bad.append((nm, nb))
    print(tag, bad)
    return bool(bad)
check('init')
nstep = 0
last = [None]
def hook(st, six):
    if last[0] is not None:
        check(f'  after {last[0]}')
    last[0] = st
for it in range(nsteps):
    nstep = dyncore_step(be, nstep, c.params['baclin'], hook=hook)
    check(f'  after {last[0]}'); last[0]=None
    u = be.get('u'); w = np.abs(u)<1e30
    print('step', nstep, 'umax', np.nanmax(np.abs(u[w])))
