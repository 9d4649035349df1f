#!/bin/bash
# the seam after a change: its tests, a minute and a half of fuzz_seam, three warm calls + a traced one, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T=${1:-r07_seam}
python -m pytest tests/test_multi_gpu.py tests/test_cli.py tests/test_configs_gpu.py -m gpu -x -q -k "not whole_suite" 2>&1 | tail -5 > gpurun_out/${T}_pytest.txt
python scripts/fuzz_seam.py ${FUZZ_S:-90} 777 2>&1 | tail -4 > gpurun_out/${T}_fuzz.txt
bash scripts/dev/e2e_trace.sh > gpurun_out/${T}_e2e.txt 2>&1
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -3 gpurun_out/${T}_pytest.txt; cat gpurun_out/${T}_fuzz.txt; cat gpurun_out/${T}_e2e.txt
