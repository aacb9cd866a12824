// Precision RAILS_PRECISION_F16X1: the f16 scoring kernels with ONE product per block (hi * hi) -- see mol_score_f16_unit.h.
#define RAILS_F16_SINGLE 1
#include "mol_score_f16.hip"
