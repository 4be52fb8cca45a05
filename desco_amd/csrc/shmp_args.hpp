// Arguments of the fused SHMP layer kernels (shmp_layer.hip: 32-row wave tiles; shmp_layer16.hip: 16-row
// wave tiles, 16 waves per CU).
#pragma once
#include <stdint.h>

namespace desco {

struct ShmpArgs {
  const float* x;
  int64_t ldx;
  const int32_t* vrowptr;
  const int32_t* vcol;
  int64_t row0, num_rows;
  int S, sm, st;
  const float* wt;          // f32 mode: [(sm+1)*64][64]
  const short* wplanes;     // x6 mode: [3][64 n][(sm+1)*64 k] bf16 planes (hi, mid, lo); f16x3 mode: [2][64 n][..] fp16 (hi, lo)
  const float* wscale;      // f16x3 mode: device {scale, 1/scale} of the weight planes (desco_split_f16x2_f32); else null
  const float* bias;
  const float* ytab;
  int64_t ldy, ytab_row0;
  float* out;
  int64_t ldo;
  float* out2;              // optional second copy of the output rows (row i - row0 of a [num_rows, *] view)
  int64_t ldo2;
  float* row_absmax;        // optional (with out2): row_absmax[i - row0] = max(row_absmax[i - row0], max_c |out[i, c]|) -- the
                            // per-row bound desco_gemm_f16x3_f32 wants for the operand these rows are a column block of
  int act;                  // DESCO_ACT_* of the epilogue (relu for the SHMP layer)
  float slope;
  // fused pooling (global_add_pool of the produced rows, gnn_model.py:107), optional: see
  // desco_shmp_layer_pool_bf16x6_f32 in desco_hip.h.  out may then be null (rows not stored).
  const uint32_t* pool_bits;   // [ceil(rows / TR)] bit r of word t: row TR t + r is the last row of its segment
  const int32_t* pool_slot;    // [ceil(rows / TR)] first partial slot of TR-row tile t
  float* pool_part;            // [num slots][64] partial segment sums
  int pool_rows;               // TR = 32 (shmp_layer.hip) or 16 (shmp_layer16.hip)
  // round 6 (16-row form only): the launch's OWN rows (the self block's operand) read from another tensor than the gather
  // sources: row i of the launch at xself + i * ldxs (i = the same global row index that addresses x).  The canonical
  // launches use it to read their rows from the anchor operand's column block (row stride 576) -- the only place the
  // canonical rows are stored since -- and pass out = NULL (rows go to out2 alone).  NULL: the rows come from x.
  const float* xself;
  int64_t ldxs;
  // ... or not read at all: with self_coef [S + 1][64] the launch's own row i is RECOMPUTED from its S slot degrees d_s (the
  // tile's row pointers are in LDS anyway) as relu(coef[S] + sum_s d_s coef[s]) -- the closed-form first layer
  // (desco_degree_affine_f32's arithmetic, bit for bit).  With x = a table of the distinct rows of that layer's output and
  // column ids that address the table, X_1 [N, 64] need not exist (gnn_model.FIRST_LAYER_TABLE).
  const float* self_coef;
};

// 16-row-tile form (shmp_layer16.hip); returns false when the shape is not one it is built for
bool shmp16_launch(const ShmpArgs& g, int cus, void* stream);

}  // namespace desco
