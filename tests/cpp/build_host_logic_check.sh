#!/bin/bash
# Builds tests/cpp/host_logic_check: the library's host sources (host half only: --cuda-host-only) + the HIP runtime stub,
# with the given sanitizer.  usage: build_host_logic_check.sh <thread|address,undefined> <out dir>
set -e
san=$1; out=$2
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/pim-embedding-lookup_amd/csrc
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
flags="-O1 -g -std=c++17 -fPIC -fno-omit-frame-pointer -I$root/include -I$src -x hip --cuda-host-only -fsanitize=$san -Wno-option-ignored -Wno-unused-command-line-argument"
mkdir -p "$out"
pids=()
for f in pimemb_kernels.hip pimemb_engine.cpp pimemb_compat.cpp pimemb_shard.cpp pimemb_peer.cpp; do      # (not pimemb_comm.cpp, the RCCL binding: the check brings a stand-in for emb_comm)
    $HIPCC $flags -c "$src/$f" -o "$out/${f%.*}.o" & pids+=($!)
done
$HIPCC $flags -c "$root/tests/cpp/hip_runtime_stub.cpp" -o "$out/hip_runtime_stub.o" & pids+=($!)
$HIPCC $flags -c "$root/tests/cpp/host_logic_check.cpp" -o "$out/host_logic_check.o" & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
# plain clang++ link: no HIP runtime, no device code (the one object with kernels refers to its device binary by a hashed name:
# point that at the stub's dummy)
nm=$(command -v nm)
defs=""
for sym in $($nm -u "$out/pimemb_kernels.o" | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u); do defs="$defs -Wl,--defsym=$sym=pimemb_stub_fatbin"; done
${HIPCC%hipcc}../lib/llvm/bin/clang++ -fsanitize=$san "$out"/*.o -o "$out/host_logic_check" $defs -lpthread -ldl -lrt
