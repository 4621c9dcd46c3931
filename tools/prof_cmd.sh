#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace (+ optional counter passes, each in its own run) of one python tool.
#   usage: tools/prof_cmd.sh <tag> <trace|fetch|write|sq|sq2 ...> -- <script.py> [args]
# rocprofv3 gets the program itself after "--" (python3 <script>), never a shell or env wrapper.
set +e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
WHAT=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do WHAT+=("$1"); shift; done
shift
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
# the script path is given relative to the repo (see usage); rocprofv3 runs from /tmp, so make it absolute first
SCRIPT=$1; shift
case $SCRIPT in /*) ;; *) SCRIPT=$REPO/$SCRIPT ;; esac
[ -f "$SCRIPT" ] || { echo "prof_cmd.sh: no such script: $SCRIPT" >&2; exit 2; }
set -- "$SCRIPT" "$@"
FAILED=0
cd /tmp && export TMPDIR=/tmp
for w in "${WHAT[@]}"; do
rc=0
case $w in
trace) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2> $OUT/trace.err; rc=$? ;;
fetch) rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 "$@" > $OUT/fetch.log 2> $OUT/fetch.err; rc=$? ;;
write) rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 "$@" > $OUT/write.log 2> $OUT/write.err; rc=$? ;;
sq)    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 "$@" > $OUT/sq.log 2> $OUT/sq.err; rc=$? ;;
sq2)   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 "$@" > $OUT/sq2.log 2> $OUT/sq2.err; rc=$? ;;
*) echo "prof_cmd.sh: unknown pass '$w'" >&2; rc=2 ;;
esac
if [ $rc -ne 0 ]; then
  # a failed pass must not look like an empty profile: say so, show the end of its stderr, skip its summary, fail the script
  echo "prof_cmd.sh: pass '$w' FAILED (exit $rc)" >&2; tail -n 15 $OUT/$w.err >&2; FAILED=1
  [ $rc -ge 124 ] && { echo "prof_cmd.sh: stopping after a killed GPU step" >&2; break; }
else
  echo "$w done"
fi
done
cd $REPO
for w in "${WHAT[@]}"; do
  d=$OUT/$w; [ "$w" = fetch ] && d=$OUT/pmc_fetch; [ "$w" = write ] && d=$OUT/pmc_write; [ "$w" = sq ] && d=$OUT/pmc_sq; [ "$w" = sq2 ] && d=$OUT/pmc_sq2
  python3 tools/summarize_prof.py $d ${TAG}_$w > $OUT/summary_$w.txt 2>&1
  cp profiles/${TAG}_$w.json $OUT/ 2>/dev/null
done
find $OUT -name "*.csv" -size +20M -delete
[ $FAILED -eq 0 ] || { echo "prof_cmd.sh: at least one pass failed" >&2; du -sh $OUT; exit 1; }
find $OUT -type f \( -name "*.db" -o -name "*.rocpd" \) -delete
du -sh $OUT
