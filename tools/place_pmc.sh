#!/bin/bash
# correlate the run-to-run spread of the scatter time with address-translation / L2 stall counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/pp_$i
  SB_QUICK=1 rocprofv3 --pmc $1 --output-format csv -d /tmp/pp_$i -- $R/tools/scatter_bench 28 > /tmp/pp_$i.log 2>&1
  f=$(find /tmp/pp_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'radix_scatter' in r['Kernel_Name'] and 'ELb1' not in r['Kernel_Name']]
by=collections.defaultdict(list)
for r in rows:
    by[r['Counter_Name']].append(float(r['Counter_Value']))
dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows if 'End_Timestamp' in r]
print('dur_ms(min)=%.3f'%min(dur) if dur else 'no ts', ' '.join('%s=%.4g'%(k,sum(v)/len(v)) for k,v in by.items()))
PY
done
