#!/bin/bash
# A/B instrument: every hardware-rate transcendental of the tile kernels replaced by the correctly rounded
# library function -- v_exp_f32 / v_log_f32 / v_sin_f32 / v_cos_f32 / v_rcp_f32 -> exp2f / log2f / sinf / cosf / IEEE division.
# Slower; numerics only.  -> variants/lib_libm.so
cd "$(dirname "$0")/../.."
python tools/build_variants.py 'libm=sed:*:s/__builtin_amdgcn_exp2f\(/exp2f(/g; s/__builtin_amdgcn_logf\(/log2f(/g; s/__sinf\(/sinf(/g; s/__cosf\(/cosf(/g; s/__expf\(/expf(/g; s/__logf\(/logf(/g; s/__builtin_amdgcn_rcpf\(/__frcp_rn(/g'
