set -o pipefail
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3z; mkdir -p $O
chk() { if grep -q "Memory access fault\|GPU coredump" "$@" 2>/dev/null; then echo "GPU FAULT: stopping"; exit 3; fi; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "known or random_small or synth1024 or random_grids" > $O/t0.log 2>&1; rc=$?; tail -4 $O/t0.log; chk $O/t0.log; if [ $rc -ne 0 ]; then exit 4; fi
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/t1.log 2>&1; rc=$?; tail -4 $O/t1.log; chk $O/t1.log; if [ $rc -ne 0 ]; then exit 5; fi
for w in c2 c4shard c5 c2h1; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>$O/err_$w.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1))" || exit 6
  chk $O/err_$w.log
done
echo "--- 9206 alone, no prof"; FX_QIDS=9206 timeout 300 python tools/gpu_prof.py 10000 --noprof 2>&1 | tail -2
echo "--- 9206 alone, prof"; FX_QIDS=9206 timeout 300 python tools/gpu_prof.py 10000 2>&1 | tail -14 | head -4
