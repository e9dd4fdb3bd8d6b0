# Board power and shader clock while the gan_step bench runs (rocm-smi polled every 0.5 s)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$\|====" | head -20
( for v in "$@"; do export "$v"; done
  timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 > gpurun_out/power_bench.log 2>/dev/null ) &
pid=$!
: > gpurun_out/power_trace.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | tr '\n' ' ' >> gpurun_out/power_trace.txt
  echo >> gpurun_out/power_trace.txt
  sleep 0.5
done
wait $pid
tail -1 gpurun_out/power_bench.log | cut -c1-200
python - <<'PY'
import re
rows = [l for l in open('gpurun_out/power_trace.txt') if l.strip()]
print(len(rows), 'samples; first / middle / last:')
for l in (rows[0], rows[len(rows)//2], rows[-3] if len(rows) > 3 else rows[-1]):
  print('  ', re.sub(r'\s+', ' ', l)[:260])
pw = [float(m.group(1)) for l in rows for m in [re.search(r'Power[^:]*:\s*([0-9.]+)', l)] if m]
ck = [float(m.group(1)) for l in rows for m in [re.search(r'sclk[^(]*\((\d+)Mhz\)', l)] if m]
if pw: print('power W: max %.0f, mean of top half %.0f' % (max(pw), sum(sorted(pw)[len(pw)//2:]) / max(1, len(pw) - len(pw)//2)))
if ck: print('sclk MHz: min %.0f max %.0f mean of lower half %.0f' % (min(ck), max(ck), sum(sorted(ck)[:max(1, len(ck)//2)]) / max(1, len(ck)//2)))
PY
