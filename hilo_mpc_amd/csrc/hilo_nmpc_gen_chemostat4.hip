// General NMPC instantiations (hilo_nmpc_gen.h) for the chemostat: hard and soft nonlinear stage constraints.
#include "hilo_nmpc_gen.h"

namespace hilo {

const GenVariant* nmpc_gen_variants_chemostat4(int* n) {
  static const GenVariant v[] = {
      gen_variant<Chemostat4, 0, 0, 2>(HILO_MODEL_CHEMOSTAT4),  // up to two hard rows
      gen_variant<Chemostat4, 0, 1, 2>(HILO_MODEL_CHEMOSTAT4),  // one soft constraint (two rows, one shared slack)
  };
  *n = (int)(sizeof(v) / sizeof(v[0]));
  return v;
}

}  // namespace hilo
