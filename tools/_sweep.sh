for wv in 2048 2560 3072 3584 4096; do echo "waves $wv: $(FXJPS_WAVES=$wv timeout -k 10 300 python bench.py --workload c2 --steps 4 --warmup 1 --no-cpu-baseline | python3 -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), round(d['ms_per_step'],1))")"; done
FXJPS_WAVES=3072 timeout -k 10 200 python tools/qstat.py c2 2>/dev/null | grep -E "longest|kernel" | head -4
