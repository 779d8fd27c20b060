# One streaming block (32 streams x 28 frames, BASELINE configs[2]) as a kernel timeline: run on the GPU box, prints to stdout
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pst && rocprofv3 --kernel-trace -d /tmp/pst -o run -- python $GRAFT_REPO_ROOT/bench_stream.py --blocks 60 --warmup 10 --no-graph > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import sqlite3, glob, re
db=glob.glob('/tmp/pst/**/*.db', recursive=True)[0]
c=sqlite3.connect(db)
rows=c.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if 'stft_fft_kernel' in r[0]]
a,b=idx[-3],idx[-2]
t0=rows[a][1]; be=t0
for name,s,e in rows[a:b]:
    nm=re.sub(r"\(.*$","",name.replace("void tvc::","").replace("tvc::",""))[:70]
    print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f} gap {max(0,s-be)/1e3:5.1f} {nm}")
    be=max(be,e)
print("# block", (rows[b][1]-t0)/1e3, "us", b-a, "kernels")
PY
