#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_tests; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -v 2>&1 | grep -v "socket.cpp" > $O/tests_full.log
grep -n "PASSED\|FAILED\|ERROR" $O/tests_full.log | tail -3 | cut -c1-200
grep -n -i "fault\|Fatal\|Abort\|core" $O/tests_full.log | head -10 | cut -c1-300
grep -n "tests/test_" $O/tests_full.log | grep -v "PASSED\|SKIPPED" | tail -12 | cut -c1-250
tail -3 $O/tests_full.log | cut -c1-200
