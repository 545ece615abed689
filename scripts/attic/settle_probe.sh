for s in 0.25 0.5 1.0 2.0; do for i in 1 2 3; do
MRX_BENCH_SETTLE_S=$s python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('settle', d['settle_s'], 'ms_per_step %.3f us kernel %.3f us' % (d['ms_per_step']*1000, d['roofline']['kernel_us']))"
done; done
