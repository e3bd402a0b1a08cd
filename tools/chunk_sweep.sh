for ch in 0 3 4 6 8 12 18 27; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt momtum_chunk=$ch 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('chunk', $ch, 'ms/step', round(d['ms_per_step'],3), 'momtum', round(d['stages_ms']['momtum'],3), 'crc', d['config']['state_crc'])"
done
