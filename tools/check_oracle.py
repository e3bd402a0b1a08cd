"""Dev helper: step the reference through the dyncore sequence and, stage by stage, feed the
same inputs to the C restatement and compare every field exactly."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from oracle.refblom import RefBackend
from oracle.coracle import COracle
from parity import copy_state, diff_report, fmt_report
cfg = sys.argv[1] if len(sys.argv) > 1 else 'chan_s'
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
c = make_case(cfg)
ref = RefBackend(cfg, c.depth)
hostinit.init_state(ref, c)
co = COracle(c.idm, c.jdm, c.kdm, ref.ntr, ref.nreg, ref.masks)
for nm, v in c.params.items():
    if not nm.endswith('0'): co.set(nm, v)
nstep = 0
state = {}
def hook(st, six):
    if 'pending' in state:
        pst, psix = state.pop('pending')
        bad = diff_report(ref, co)
        print(f'step {nstep+1} {pst}:', 'OK' if not bad else 'MISMATCH\n' + fmt_report(bad))
    if only and st not in only: return
    try:
        copy_state(ref, co)
        co.set('nstep', nstep + 1); co.set('delt1', ref.ref.get_real('delt1'))
        co.stage(st, *six)
        state['pending'] = (st, six)
    except KeyError as e:
        print('skip', e)
for it in range(nsteps):
    nstep_new = dyncore_step(ref, nstep, c.params['baclin'], hook=hook)
    if 'pending' in state:
        pst, _ = state.pop('pending')
        bad = diff_report(ref, co)
        print(f'step {nstep+1} {pst}:', 'OK' if not bad else 'MISMATCH\n' + fmt_report(bad))
    nstep = nstep_new
