R=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_s1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s1 -- python $R/bench.py --stage 1 --no-cpu-baseline --no-prof --steps 60 --warmup 20 > /tmp/s1.json 2>/tmp/s1.err
python - <<'PY'
import csv, sys, glob, collections, re
f=glob.glob('/tmp/prof_s1/**/*kernel_trace.csv', recursive=True)[0]
rows=[]
for r in csv.reader(open(f)):
    if len(r)>=11 and r[9].isdigit(): rows.append((int(r[9]),int(r[10]),r[7]))
rows.sort()
heads=[i for i,r in enumerate(rows) if 'stage1_head_kernel' in r[2]]
a,b=heads[-2],heads[-1]
t0=rows[a][1]
for r in rows[a+1:b+1]:
    n=re.sub(r'at::native::','',r[2])
    n=re.sub(r'\(anonymous namespace\)::','',n)
    print(f"{(r[0]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:7.1f}  {n[:150]}")
PY
