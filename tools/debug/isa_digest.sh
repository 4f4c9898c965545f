#!/bin/bash
# usage: tools/debug/isa_digest.sh OUTDIR
# Device-only assembly (hipcc -S --cuda-device-only) of every translation unit under the four product flag sets, and a digest
# per file with the compilation-unit id (__hip_cuid_<hash of the source text>) masked: two source trees whose digests agree
# compile to the same instructions.  Used for profiles/r05_switch_cleanup.txt (removing preprocessor switches must not move
# an instruction).
out=$(realpath -m $1); mkdir -p $out
cd "$(dirname "$0")/../../packppi_amd/csrc"
B="-O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -DPP_BUILD_ID=\"x\" -w"
gen() { tag=$1; shift; flags=$1; shift; for s in "$@"; do /opt/rocm/bin/hipcc $B $flags $s -o $out/${s%.hip}.$tag.s & done; wait; }
gen def "-DPP_EDGE_F16" pp_api.hip pp_prepare.hip pp_node.hip pp_edge_f16.hip pp_clash.hip
gen f32 "" pp_api.hip pp_prepare.hip pp_node.hip pp_edge.hip pp_clash.hip
gen chk "-DPP_EDGE_F16 -DPP_CHECK_RANGE" pp_api.hip pp_prepare.hip pp_node.hip pp_edge_f16.hip pp_clash.hip
gen dbg "-DPP_EDGE_F16 -DPP_DIAG" pp_api.hip pp_prepare.hip pp_node.hip pp_edge_f16.hip pp_clash.hip
(cd $out && for f in *.s; do echo "$(sed -E "s/__hip_cuid_[0-9a-f]+/__hip_cuid_X/g" $f | sha256sum | cut -c1-16)  $(grep -c . $f) lines  $f"; done > DIGEST)
cat $out/DIGEST
