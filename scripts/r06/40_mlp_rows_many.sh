#!/bin/bash
# Round 6: k_mlp_rows_many (up to 16 batches per launch) -- parity, then config 5 and EmbeddingMLP.py literal: strict, several batches per launch, and the
# two-streams form it replaces (SPRK_MLP_ROWS_MANY=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_43}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stated_sizes.py tests/test_gpu_host_api.py -m gpu -x -q -k "mlp or wide or embedding or every_tile or many or predict" 2>&1 | tail -3 | tee $O/pytest.txt
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('value %.3f G/s | %.2f us/step (%s batches per launch) | strict %.2f us' % (l['value']/1e9, 1e3*l['ms_per_step'], l['config'].get('batches_per_launch'), r['avg_launch_us']))"; }
for rep in 1 2; do
  for v in 1 0; do
    for w in widedeep_c5 embedding_mlp_ref; do
      echo "SPRK_MLP_ROWS_MANY=$v $w: $(SPRK_MLP_ROWS_MANY=$v timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe --variants 0 2>>$O/err.txt | tail -1 | tee -a $O/lines.jsonl | get)" | tee -a $O/timing.txt
    done
  done
done
