// Shapes beyond the four tuned ones (8x4x64, 8x4x128, 8x8x32, 16x16x64 with H = 128): every (P_Q, P_X, d, H) in this list is
// built as the "direct" shell (independent waves, 8 per workgroup) in both precisions.  The register-resident scheme admits any
// combination with P_Q in {8, 16, 32}, L = P_Q * P_X in {32, 64}, d % 16 == 0, H % 64 == 0, H <= 128; the list is what is
// instantiated (each entry costs two kernels of compile time).  X(P_Q, P_X, d, H)
#pragma once
#define MOL_EXTRA_SHAPES(X) \
  X(8, 4, 32, 128)          \
  X(8, 8, 16, 128)          \
  X(8, 8, 64, 128)          \
  X(8, 8, 128, 128)         \
  X(8, 8, 48, 128)          \
  X(8, 4, 16, 128)          \
  X(16, 2, 64, 128)         \
  X(16, 4, 32, 128)         \
  X(16, 4, 64, 128)         \
  X(32, 2, 32, 128)         \
  X(8, 8, 32, 64)           \
  X(8, 4, 64, 64)

// Pair gate WITHOUT hidden layer (gating_qi_hidden_dim <= 0, modeling/similarity_utils.py:199-206): exact-fp32 precision, direct
// shell.  X(P_Q, P_X, d)
#define MOL_NOHID_SHAPES(X) \
  X(8, 8, 32)               \
  X(8, 4, 64)               \
  X(16, 4, 32)
