set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04x
( time bash scripts/profile_round.sh r04 ) > gpurun_out/r04x/profile_round.log 2>&1
tail -5 gpurun_out/r04x/profile_round.log; ls gpurun_out/r04 | wc -l
