#!/bin/bash
# Evidence for the checks `sortedArraysHold` of tests/cpp/headless_tick.cpp (what prepareSortedMeshes / sortMeshes leave behind, against
# the reference TEXT): the same headless_tick built against copies of the drop-in shim that are broken on purpose, one statement each.
# Every broken shim must FAIL with the check's own message ("GPU system vs the text of mesh.cpp..."), the shim as it is passes.
# Run on a GPU box from the repository root; prints one block per case.
set -u
root=$(pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/tests/cpp" "$tmp/garden_amd/csrc" "$tmp/oracle" "$tmp/include"
cp "$root/tests/cpp/headless_tick.cpp" "$tmp/tests/cpp/"
cp "$root"/oracle/*.hpp "$root"/oracle/*.h "$root"/oracle/*.c "$tmp/oracle/"
cp "$root/include/garden_vis.h" "$tmp/include/"
cd "$tmp/tests/cpp"
gcc -O2 -march=haswell -ffp-contract=off -fno-fast-math -std=c11 -pthread -c ../../oracle/gv_oracle.c -o gv_oracle.o
gcc -O2 -march=haswell -ffp-contract=off -fno-fast-math -std=c11 -pthread -c ../../oracle/gv_oracle_avx2.c -o gv_oracle_avx2.o
build() {  # build <binary> <sed expression>: the shim with one statement broken
    rm -rf "$tmp/garden_amd/csrc/host"; cp -r "$root/garden_amd/csrc/host" "$tmp/garden_amd/csrc/"
    before=$(md5sum < "$tmp/garden_amd/csrc/host/gpu_visibility_system.hpp")
    sed -i -e "$2" "$tmp/garden_amd/csrc/host/gpu_visibility_system.hpp"
    [ "$before" != "$(md5sum < "$tmp/garden_amd/csrc/host/gpu_visibility_system.hpp")" ] || { echo "the patch did not apply: $2"; exit 2; }
    g++ -O2 -std=c++17 -Wno-invalid-offsetof -fno-strict-aliasing -march=haswell -ffp-contract=off -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include headless_tick.cpp gv_oracle.o gv_oracle_avx2.o \
        -o "$1" -L"$root/garden_amd/lib" -lgarden_vis -Wl,-rpath,"$root/garden_amd/lib" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -lm -lpthread || exit 2
}
args="--mode both --entities 30000 --mixed --csm --hier --ticks 2"
echo "== the shim as it is: $args =="
"$root/tests/cpp/build/headless_tick" $args; echo "exit code $?"
echo "== (i) mesh.cpp:252,419-421 — the records of a sorted system carry ANOTHER system's bufferIndex (bufferIndex ^ 1 in the record layout) =="
build tick_i 's/(uint32_t)offsetof(SortedMesh, bufferIndex), sp.bufferIndex)/(uint32_t)offsetof(SortedMesh, bufferIndex), sp.bufferIndex ^ 1u)/'
./tick_i $args; echo "exit code $?"
echo "== (ii) mesh.cpp:416-419 — a shadow pass's records keep the LIGHT pass's bufferIndex, which counts the UI system (the fix-up taken out) =="
build tick_ii 's/if (shadowIndex != bufferIndex)  \/\/ (records built on the device carry/if (false)  \/\/ (records built on the device carry/'
./tick_ii $args; echo "exit code $?"
echo "== (iii) mesh.cpp:270-295, mesh.hpp:196,204 — sort directions swapped (unsorted buffers descending, sorted arrays ascending) =="
build tick_iii 's/gv_pool_sort(ctx, p, v, sp.sorted ? 1 : 0)/gv_pool_sort(ctx, p, v, sp.sorted ? 0 : 1)/'
./tick_iii $args; echo "exit code $?"
echo "== (iv) mesh.cpp:255-259 — a sorted buffer's drawCount one above the records it appended =="
build tick_iv 's/counters->drawCount = r.draw_count;/counters->drawCount = r.draw_count + 1;/'
./tick_iv $args; echo "exit code $?"
rm -rf "$tmp"
