// Precision RAILS_PRECISION_F16X1 for the shapes of mol_score_extra_shapes.h -- see mol_score_f16_unit.h.
#define RAILS_F16_SINGLE 1
#include "mol_score_f16_extra.hip"
