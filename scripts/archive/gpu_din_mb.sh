timeout 300 python -m pytest tests -m gpu -q -x -k "several_batches_per_launch_din" 2>&1 | tail -2
echo "attn MB on:"; SPRK_DIN_ATTN_MB=1 timeout 300 python -m pytest tests -m gpu -q -x -k "several_batches_per_launch_din" 2>&1 | tail -2
for cfg in "16 2" "8 2" "8 0"; do set -- $cfg; echo "din lb=$1 streams=$2 $(python bench.py --steps 320 --warmup 32 --workload din_c3 --cpu-seconds 0 --launch-batches $1 --overlap-streams $2 2>&1 | tail -1 | grep -o "\"ms_per_step\": [0-9.]*\|\"value\": [0-9.]*" | tr "\n" " ")"; done
