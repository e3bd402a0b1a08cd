import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_stage_parity as T
cfg = sys.argv[1]; n = int(sys.argv[2])
try:
    T._run(cfg, n, T.GPU_STAGES)
    print(cfg, n, "steps: every stage within its stated tolerance (exact unless exp() is involved)")
except AssertionError as e:
    print(str(e)[:3000])
