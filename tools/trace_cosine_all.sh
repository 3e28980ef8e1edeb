set -o pipefail
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/costr; rm -rf $O; mkdir -p $O
for q in 16 32 48 64; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t$q -- python3 tools/bench_cosine.py --nq $q --shapes 1000000x768 > $O/b$q.json 2> $O/e$q.err && python3 tools/trace_cosine_chain.py $O/t$q > $O/cosine_chain_${q}q.txt
  rm -rf $O/t$q
done
cat $O/cosine_chain_*q.txt
