cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python $R/bench.py --no-cpu-baseline --repeats 2 --steps 10 --channels 8 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pf/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE': acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in acc.items():
    if len(v) > 5: print(k, len(v), 'mean FETCH_SIZE KiB', sum(v)/len(v), '-> x2 corrected MB', 2*sum(v)/len(v)*1024/1e6)
PY
