#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum --output-format csv -d gpurun_out/${TAG:-ta_pmc} -o t -- ${MICRO:-./scripts/micro/ta_cost} > /dev/null 2>&1
TAG=${TAG:-ta_pmc} python3 - <<'PY'
import csv,glob,collections,os
rows=collections.OrderedDict()
for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % os.environ["TAG"], recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_loads" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]),{})[r["Counter_Name"]]=float(r["Counter_Value"])
for d,v in rows.items(): print(d, {k:int(x) for k,x in v.items()})
PY
