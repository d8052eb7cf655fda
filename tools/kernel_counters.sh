#!/bin/bash
# One rocprofv3 --pmc pass over a short S3 run, printing the given counters per dispatch group of the kernels whose name
# contains <substring> (launches grouped by grid size, so that e.g. the permanent and the induced spread stay apart):
#   tools/kernel_counters.sh <outname> <substring> COUNTER [COUNTER ...]        (env ADMP_HIP_LIB, WORKLOAD optional)
out=$1; sub=$2; shift 2
root=$(pwd); export TMPDIR=/tmp
rm -rf gpurun_out/$out
timeout -k 5 120 rocprofv3 --pmc "$@" --kernel-trace -d $root/gpurun_out/$out --output-format csv -- python3 $root/bench.py --workload ${WORKLOAD:-S3} --steps 3 --warmup 2 --no-cpu --no-scale --no-extras > gpurun_out/$out.log 2>&1
python3 - "$out" "$sub" <<'PY'
import csv, glob, collections, sys
out, sub = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob('gpurun_out/%s/**/*kernel_trace.csv' % out, recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv' % out, recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            key = (r['Kernel_Name'].replace('void admp::', '')[:48], r['Grid_Size'])
            per[key][r['Counter_Name']].append(float(r['Counter_Value']))
            per[key]['us'].append(dur.get(r['Dispatch_Id'], 0.0))
for k, d in per.items():
    print(k[0], 'grid', k[1], 'launches', len(d['us']) // max(1, len(d) - 1))
    for c, v in sorted(d.items()):
        print('   %-28s %14.1f' % (c, sum(v) / len(v)))
PY
