set -e
for few in 40 8; do for g in 4 8 16 64; do echo "FEW=$few GROWTH=$g"; UCFP_HAMMING_FEW=$few UCFP_HAMMING_GROWTH_SMALL=$g python tools/bench_hamming.py --n 12500000 --nq 9 16 32 64 128 --reps 30 2>/dev/null | python -c "
import sys, json
print(' '.join(f\"{json.loads(l)['nq']}q:{json.loads(l)['ms']*1000:.0f}us\" for l in sys.stdin if l.startswith('{')))"; done; done
