cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tr50
rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d /tmp/tr50 -- python /root/repo/scripts/host_path_rate.py 50000 > /tmp/tr50.log 2>&1
tail -2 /tmp/tr50.log
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('/tmp/tr50/**/*.csv', recursive=True):
    kind = 'kernel' if 'kernel_trace' in f else 'memcpy' if 'memory_copy' in f else 'hip' if 'hip_api' in f else None
    if not kind: continue
    for r in csv.DictReader(open(f)):
        s=int(r.get('Start_Timestamp',0)); e=int(r.get('End_Timestamp',0))
        name = r.get('Kernel_Name') or r.get('Name') or r.get('Function') or r.get('Direction') or ''
        rows.append((s,e,kind,name[:60], r.get('Size','') ))
rows.sort()
# last pg_search_batch call: find the last hipMemsetAsync burst... just print the last 90 events
t0=rows[-int(__import__("os").environ.get("TR_N","90"))][0]
for s,e,k,n,sz in rows[-int(__import__("os").environ.get("TR_N","90")):]:
    print(f"{(s-t0)/1000:9.1f} us  +{(e-s)/1000:8.1f} us  {k:7s} {n} {sz}")
PY
