#!/bin/bash
# probe tables of the old and the new build, bit for bit (development aid)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_old.so python3 scripts/dev/table_hashes.py > gpurun_out/tab_old.txt 2>gpurun_out/tab_old.err
python3 scripts/dev/table_hashes.py > gpurun_out/tab_new.txt 2>gpurun_out/tab_new.err
tail -n 2 gpurun_out/tab_old.err gpurun_out/tab_new.err
wc -l gpurun_out/tab_old.txt gpurun_out/tab_new.txt
diff gpurun_out/tab_old.txt gpurun_out/tab_new.txt && echo TABLES IDENTICAL
