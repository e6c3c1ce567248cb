R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for CFG in ragged chr20; do
  for DT in i16 f32; do
  for O in 0 10 50; do
    for OFF in 0 1; do
      NMOD_NO_COUNTING=$OFF python3 $R/bench.py --config $CFG --positions 2000000 --dtype $DT --spread 200 --outliers $O --steps 5 --warmup 2 --no-cpu --no-side --no-host-path --side-file /tmp/sweep_side.json 2>/tmp/sweep_err.txt | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KS %-8s %s outliers %3d permille counting=%s  %.4g pos/s  K1 %.3f ms  verify %s  %s' % ('$CFG', '$DT', $O, 'off' if $OFF else 'on ', d['value'], d['roofline']['kernel_avg_ms'], d['verify']['ok'], d['form_share']))" || tail -5 /tmp/sweep_err.txt
    done
  done
  done
done
