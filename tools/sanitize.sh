#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the host code (planner, handles, SpinConserve layout tables,
# exchange schedules, Krylov drivers -- `python -m dynamite_amd.build --sanitize`: sanitizers on the host side of every
# source) and of the C oracle, under the CPU tests that drive them through host-only handles.  CPU container only.
#   tools/sanitize.sh [pytest args]      -> /tmp/sanitize.log (pytest) and /tmp/sanitize_reports.* (the sanitizers)
set -u
cd "$(dirname "$0")/.."
export DNM_LIB_VARIANT=san DNM_EXPERIMENTAL=1
python3 -m dynamite_amd.build --sanitize > /dev/null || exit 1
make -C oracle -s libdnm_oracle_san.so || exit 1
RT=$(python3 -c "from dynamite_amd import build; print(build.sanitizer_runtime())")
OMP=/opt/rocm/lib/llvm/lib/libomp.so
rm -f /tmp/sanitize_reports.*
# Python itself is not instrumented: its interpreter-lifetime allocations are not leaks of ours (detect_leaks=0).  ASan
# errors are fatal; UBSan reports and carries on (one run lists every finding).  Reports go to files: pytest captures
# stderr, and a process ASan ends takes the captured text with it.
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:strict_string_checks=1:detect_stack_use_after_return=1:check_initialization_order=1:log_path=/tmp/sanitize_reports
export UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/sanitize_reports
TESTS=${@:-tests/test_host_logic.py tests/test_sc3_layout.py tests/test_transpose_exchange.py tests/test_distributed_gloo.py tests/test_oracle.py tests/test_abi.py tests/test_bench_launch.py tests/test_sanitized_sizes.py tests/test_fake_rccl.py}
LD_PRELOAD="$RT:$OMP" python3 -m pytest $TESTS -q -m "not gpu" -p no:cacheprovider 2>&1 | tee /tmp/sanitize.log
echo "sanitizer reports: $(cat /tmp/sanitize_reports.* 2>/dev/null | grep -c 'runtime error\|ERROR: AddressSanitizer')" | tee -a /tmp/sanitize.log
cat /tmp/sanitize_reports.* 2>/dev/null | grep 'runtime error\|ERROR: AddressSanitizer\|SUMMARY' | sort | uniq -c | sort -rn | head -40 | tee -a /tmp/sanitize.log
