python bench.py --camera front --steps 6 --warmup 2 --cpu-frames 0 --no-streaming-pass --verify-windows 4 --optimise-only-steps 2 --closed-loop-frames 0 2>&1 | tail -1 | cut -c1-400
python bench.py --batch 64 --steps 6 --warmup 2 --cpu-frames 0 --verify-windows 2 --optimise-only-steps 2 --closed-loop-frames 0 2>&1 | tail -1 | cut -c1-300
python bench.py --ba-views random --steps 6 --warmup 2 --cpu-frames 0 --no-streaming-pass --verify-windows 4 --optimise-only-steps 0 --closed-loop-frames 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['bound'], d['config']['ba_check'])"
