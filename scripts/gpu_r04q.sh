export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ref_pin.py -q -m gpu -x 2>&1 | grep -v "^E   *+\|^E   *$" | cut -c1-300 | tail -12
timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_ref_pin.py 2>&1 | tail -3
