for v in dump7pE dump7pB dump7pF; do echo "== $v"; V=$v RUNS=7 bash scripts/r06/04_dien_dump.sh r06_05_$v 2>&1 | grep -v "^launch\|^      \|^     word\|amdgpu.ids"; done
