set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04k
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r04k/pytest.txt 2>&1
( time FN2_PROFILE_LIGHT=1 bash scripts/profile_round.sh r04 ) > gpurun_out/r04k/profile_round.log 2>&1
tail -5 gpurun_out/r04k/pytest.txt; tail -5 gpurun_out/r04k/profile_round.log; ls gpurun_out/r04 | head -50
