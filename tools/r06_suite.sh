#!/bin/bash
# round 6: the whole -m gpu suite + smoke, as the driver runs them
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_suite; mkdir -p $O
( time timeout 2400 python3 -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -30 $O/pytest.log | cut -c1-300
( time python3 -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
tail -3 $O/smoke.log
