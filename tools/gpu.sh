#!/bin/bash
# One parameterised script for every GPU-box job of a round (replaces the per-experiment one-offs):
#   gpurun -- 'bash tools/gpu.sh <task> [args] [-- <task> [args] ...]'
# tasks:
#   tests [pytest args]         GPU suite (default: whole suite), log in gpurun_out/pytest_gpu.log
#   smoke                       __graft_entry__.smoke()
#   bench [bench.py args]       default bench line -> gpurun_out/bench_default.log
#   ab 'ENV=v ENV=v' ...        A/B of the gan_step bench under env settings (baseline first), 2 reps
#   abwarp 'ENV=v' ...          the same for --workload warp (random and room depth; WARP_H=512 in the environment: at that height)
#   prof_step TAG [ENV=v ...]   rocprofv3 kernel stats of the default bench -> gpurun_out/TAG_kernel_stats.csv
#   prof_warp TAG [ENV=v ...]   kernel stats of the warp bench, random + room depth
#   pmc_conv TAG "SHAPE"        MFMA-busy / FETCH / WRITE counters of tools/one_conv.py SHAPE (3 passes)
#   pmc_warp TAG                FETCH / WRITE counters of the warp kernels (2 passes)
#   pmc_warp_valu TAG           VALU / LDS counters of the warp kernels (1 pass) -> gpurun_out/TAG_warp_valu_pmc.json
#   pmc_step TAG [ENV=v ...]    MFMA-busy + shader clock of every kernel inside the step (1 pass, tools/pmc_step.py)
#   prof_py TAG <file.py> [args] rocprofv3 kernel stats of any python tool -> gpurun_out/TAG_kernel_stats.csv
#   pmc_py TAG "PMC ..." REGEX <file.py> [args]   one --pmc pass over any python tool -> gpurun_out/TAG_pmc.json
#   py <file.py> [args]         any python tool
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
line() { python -c "import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.2f value %.3f frac %.4f' % (d['ms_per_step'], d['value'], r['frac']), 'in_step %.4f' % r['frac_in_step'] if 'frac_in_step' in r else '')"; }
task_tests() {
  SECONDS=0
  if [ $# -eq 0 ]; then set -- tests; fi
  timeout 3000 python -m pytest "$@" -m gpu -x -q --durations=10 > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$? elapsed $SECONDS s"; tail -22 gpurun_out/pytest_gpu.log | cut -c1-200
}
task_smoke() {
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
}
task_bench() {
  SECONDS=0
  timeout 1200 python bench.py "$@" > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
  echo "bench rc=$? elapsed $SECONDS s"; tail -1 gpurun_out/bench_default.log | cut -c1-7000; tail -3 gpurun_out/bench_default.err | cut -c1-300
}
task_ab() {
  for rep in 1 2; do
    for v in "SE3DS_NOP=1" "$@"; do
      echo "== $v"
      env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 2>gpurun_out/ab.err | line || tail -5 gpurun_out/ab.err
    done
  done
}
task_abwarp() {
  for rep in 1 2; do
    for v in "SE3DS_NOP=1" "$@"; do
      for d in random room; do
        echo "== $v ($d)"
        env $v timeout 300 python bench.py --workload warp --warp-height ${WARP_H:-1024} --warp-depth $d --steps 200 --warmup 20 --no-cpu-baseline 2>gpurun_out/ab.err | python -c "import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('us/render %.1f frac %.4f step_ms %.4f' % (1e3*r['ms_per_launch'], r['frac'], d['ms_per_step']))" || tail -5 gpurun_out/ab.err
      done
    done
  done
}
task_prof_step() {
  local tag=$1; shift
  rm -rf gpurun_out/prof_tmp
  local note="$* rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32   (3 warm-up + 10 timed + 2 instrumented train_g_d steps = 15 steps; model build kernels included)"
  ( for v in "$@"; do export "$v"; done
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o gan -- python bench.py --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 > gpurun_out/prof_$tag.log 2>&1 )
  python tools/rocpd_summary.py gpurun_out/prof_tmp/gan_results.db gpurun_out/${tag}_kernel_stats.csv "$note"
  python tools/step_trace.py gpurun_out/prof_tmp/gan_results.db gpurun_out/${tag}_per_step.csv 4 11 "$note"
  grep '^#' gpurun_out/${tag}_per_step.csv | cut -c1-200
  tail -1 gpurun_out/prof_$tag.log | cut -c1-400
  head -24 gpurun_out/${tag}_kernel_stats.csv | cut -c1-160; tail -1 gpurun_out/${tag}_kernel_stats.csv
  rm -rf gpurun_out/prof_tmp
}
task_prof_warp() {
  local tag=$1; shift
  for v in "$@"; do export "$v"; done
  for d in random room; do
    rm -rf gpurun_out/prof_tmp
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o warp -- python bench.py --workload warp --warp-height ${WARP_H:-1024} --warp-depth $d --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
    python tools/rocpd_summary.py gpurun_out/prof_tmp/warp_results.db gpurun_out/${tag}_warp_${d}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload warp --warp-height ${WARP_H:-1024} --warp-depth $d --steps 200 --warmup 20 --no-cpu-baseline"
    head -9 gpurun_out/${tag}_warp_${d}_kernel_stats.csv | cut -c1-150
  done
  rm -rf gpurun_out/prof_tmp
}
task_pmc_conv() {
  local tag=$1 shape="$2"
  local st=$(echo $shape | tr ' ' '_')
  rm -rf gpurun_out/pmc_tmp_*
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_tmp_a -o pmc -- python tools/one_conv.py $shape > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_tmp_f -o pmc -- python tools/one_conv.py $shape > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_tmp_w -o pmc -- python tools/one_conv.py $shape > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/${tag}_conv_pmc_$st.json --source=se3ds_amd/csrc/conv.hip "gpurun_out/pmc_tmp_a/*.db" "gpurun_out/pmc_tmp_f/*.db" "gpurun_out/pmc_tmp_w/*.db" 'igemm|wgrad'
  rm -rf gpurun_out/pmc_tmp_*
}
task_pmc_warp() {
  local tag=$1
  rm -rf gpurun_out/pmc_tmp_*
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_tmp_f -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_tmp_w -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/${tag}_warp_pmc.json --source=se3ds_amd/csrc/geom.hip "gpurun_out/pmc_tmp_f/*.db" "gpurun_out/pmc_tmp_w/*.db" 'splat|unproject'
  rm -rf gpurun_out/pmc_tmp_*
}
task_pmc_warp_valu() {
  # VALU / LDS counters of the warp kernels (one pass): is the splat VALU-bound?
  local tag=$1
  rm -rf gpurun_out/pmc_tmp_v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_tmp_v -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/${tag}_warp_valu_pmc.json --source=se3ds_amd/csrc/geom.hip "gpurun_out/pmc_tmp_v/*.db" 'splat|unproject'
  rm -rf gpurun_out/pmc_tmp_v
}
task_pmc_step() {
  local tag=$1; shift
  rm -rf gpurun_out/pmc_tmp_s
  ( for v in "$@"; do export "$v"; done
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_tmp_s -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 > gpurun_out/pmc_step_$tag.log 2>&1 )
  tail -1 gpurun_out/pmc_step_$tag.log | cut -c1-300
  python tools/pmc_step.py "gpurun_out/pmc_tmp_s/*.db" gpurun_out/${tag}_step_pmc.json 40
  rm -rf gpurun_out/pmc_tmp_s
}
task_prof_py() {
  local tag=$1; shift
  rm -rf gpurun_out/prof_tmp
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o run -- python "$@" > gpurun_out/prof_$tag.log 2>&1
  python tools/rocpd_summary.py gpurun_out/prof_tmp/run_results.db gpurun_out/${tag}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python $*"
  head -${PROF_HEAD:-16} gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
  rm -rf gpurun_out/prof_tmp
}
task_pmc_py() {
  local tag=$1 pmc="$2" re="$3"; shift; shift; shift
  rm -rf gpurun_out/pmc_tmp_g
  rocprofv3 --kernel-trace --pmc $pmc -d gpurun_out/pmc_tmp_g -o pmc -- python "$@" > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/${tag}_pmc.json "gpurun_out/pmc_tmp_g/*.db" "$re"
  python - gpurun_out/${tag}_pmc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
  if k.startswith('_'): continue
  print(k[:60], ' '.join('%s=%.4g' % (a, (b['avg'] if isinstance(b, dict) and 'avg' in b else b)) for a, b in v.items() if not isinstance(b, dict) or 'avg' in b))
PY
  rm -rf gpurun_out/pmc_tmp_g
}
task_py() { timeout 1500 python "$@"; }
args=()
run_task() { if [ ${#args[@]} -gt 0 ]; then local t=${args[0]}; echo "##### $t ${args[*]:1}"; "task_$t" "${args[@]:1}"; fi; args=(); }
for a in "$@"; do
  if [ "$a" == "--" ]; then run_task; else args+=("$a"); fi
done
run_task
du -sh gpurun_out | cut -c1-40
