#!/bin/bash
# The host layer of the library (nanorq_api.c, io.c, planner_host.cpp: the object API, its threads, the host planner) under the
# sanitizers, like the reference's own `make ... -fsanitize` targets (reference Makefile:95-99).
#   bash tools/sanitize.sh build      (build container)  build_var/libnanorq_hip_asan.so (-fsanitize=address,undefined),
#                                     build_var/libnanorq_hip_tsan.so (-fsanitize=thread) and tools/sanitize_exercise.c linked
#                                     against each; the kernels' object is the regular build's
#   bash tools/sanitize.sh cpu        (anywhere)         the CPU tier's host-logic tests on the ASan/UBSan library
#   bash tools/sanitize.sh gpu        (GPU box, gpurun)  sanitize_exercise faults on the ASan/UBSan library (every entry point that
#                                     moves bytes, with the n-th runtime call failing), the file CLI (tools/rqfile.c) built
#                                     against it, then sanitize_exercise threads on the TSan library
# Any report makes the step fail (halt_on_error / -fno-sanitize-recover); leak checking is off: the HIP runtime keeps its
# allocations for the life of the process.  Output: gpurun_out/sanitize/*.log (copied to profiles/ by hand).
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so); TSAN=$(gcc -print-file-name=libtsan.so)
OUT=$REPO/gpurun_out/sanitize; mkdir -p $OUT
case "${1:-build}" in
build)
  python - <<'PY'
import os
from nanorq_amd import build
build.build_lib()   # (the kernels' object)
root = os.path.dirname(os.path.dirname(os.path.abspath(build.__file__)))
os.makedirs(os.path.join(root, "build_var"), exist_ok=True)
build.build_lib(out=os.path.join(root, "build_var", "libnanorq_hip_asan.so"), reuse_hip_obj=True, verbose=True,
                host_extra=["-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
build.build_lib(out=os.path.join(root, "build_var", "libnanorq_hip_tsan.so"), reuse_hip_obj=True, verbose=True,
                host_extra=["-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=thread"])
PY
  gcc -std=c99 -D_DEFAULT_SOURCE -g -O1 -fsanitize=thread -Iinclude tools/sanitize_exercise.c -o build_var/exercise_tsan \
      -Lbuild_var -l:libnanorq_hip_tsan.so -Wl,-rpath,'$ORIGIN' -lpthread -lm || exit 1
  gcc -std=c99 -D_DEFAULT_SOURCE -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude tools/sanitize_exercise.c \
      -o build_var/exercise_asan -Lbuild_var -l:libnanorq_hip_asan.so -Wl,-rpath,'$ORIGIN' -lpthread -lm || exit 1
  gcc -std=c99 -D_DEFAULT_SOURCE -D_FILE_OFFSET_BITS=64 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude tools/rqfile.c \
      -o build_var/rqfile_asan -Lbuild_var -l:libnanorq_hip_asan.so -Wl,-rpath,'$ORIGIN' -lpthread -lm || exit 1
  ;;
cpu)
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
    NANORQ_HIP_LIB=$REPO/build_var/libnanorq_hip_asan.so python -m pytest tests/test_api_host.py tests/test_tables.py -x -q 2>&1 | tee $OUT/asan_cpu.log | tail -5
  ;;
gpu)
  export GPU_MAX_HW_QUEUES=8 NANORQ_HIP_FAULT_INJECT=1   # (the host process's to set: include/nanorq.h; "fail_after" is a test facility)
  export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  ( $REPO/build_var/exercise_asan faults; echo "asan exercise rc=$?"
    head -c 3000000 /dev/urandom > /tmp/san_in.bin
    $REPO/build_var/rqfile_asan encode /tmp/san_in.bin 1024 -o /tmp/san.rq -l 8 -x 3; echo "asan rqfile encode rc=$?"
    $REPO/build_var/rqfile_asan decode /tmp/san_out.bin -i /tmp/san.rq; echo "asan rqfile decode rc=$?"
    cmp /tmp/san_in.bin /tmp/san_out.bin && echo "asan rqfile round trip identical" ) 2>&1 | tee $OUT/asan_gpu.log | tail -8
  # (ThreadSanitizer needs its fixed address-space layout: no ASLR)
  TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:suppressions=$REPO/tools/tsan.supp setarch $(uname -m) -R $REPO/build_var/exercise_tsan threads 2>&1 | tee $OUT/tsan_gpu.log | tail -8
  echo "tsan rc=${PIPESTATUS[0]}" | tee -a $OUT/tsan_gpu.log
  ;;
esac
