# the GPU test tier as the driver runs it, log under gpurun_out/<tag>/pytest.txt
set -u
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
( time timeout 1800 python -m pytest tests -m gpu -q ${2:-} ) > gpurun_out/$TAG/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.txt
grep -v "^\.\|^$" gpurun_out/$TAG/pytest.txt | tail -40
