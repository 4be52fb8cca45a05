// Include BEFORE common_device.hpp (and close the translation unit with tu_no_packed_f32_end.hpp): every device function of
// the file is compiled without packed fp32 instruction selection (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  For kernels
// whose vector work runs beside MFMAs or is the bound itself: on MI355X a packed fp32 instruction costs more issue time than
// the two plain ones it replaces (same-box A/B, profiles/r6_ai_ab_no_packed_f32.log: gossip kernel 8.24 -> 8.10 ms, count
// head from the embeddings 0.70 -> 0.62 ms although its inner loop grows from 6 to 8 instructions per quad; the layer
// kernel loses 4 % -- it spills at its 168-register budget -- and keeps packed selection).  The attribute must cover the
// helpers too (a function with the feature cannot be inlined into one without), hence a pragma over the whole file; the
// device library's id queries (threadIdx / blockIdx / gridDim) would stay real calls for the same reason: such files use
// the __builtin_amdgcn_workitem_id_x / workgroup_id_x builtins and take the grid size as an argument.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#endif
