// General NMPC instantiations (hilo_nmpc_gen.h) for the mobile robot: path following, optionally with a soft stage
// constraint (SURVEY 8d configuration C5); registry of all general variants.
#include "hilo_nmpc_gen.h"

namespace hilo {

const GenVariant* nmpc_gen_variants_chemostat4(int* n);

static const GenVariant* variants_robot6(int* n) {
  static const GenVariant v[] = {
      gen_variant<Robot6, 1, 0, 0>(HILO_MODEL_ROBOT6),  // path following
      gen_variant<Robot6, 1, 1, 2>(HILO_MODEL_ROBOT6),  // path following + one soft constraint
      gen_variant<Robot6, 0, 1, 2>(HILO_MODEL_ROBOT6),  // set-point tracking + one soft constraint
      gen_variant<Robot6, 1, 1, 2, true>(HILO_MODEL_ROBOT6),  // C5 at N = 50: iterate in the global workspace
  };
  *n = (int)(sizeof(v) / sizeof(v[0]));
  return v;
}

const GenVariant* nmpc_gen_find(int model_id, int nth, int ne, int nc_needed, int N) {
  const GenVariant* (*tables[])(int*) = {variants_robot6, nmpc_gen_variants_chemostat4};
  const GenVariant* best = nullptr;
  for (auto tab : tables) {
    int n = 0;
    const GenVariant* v = tab(&n);
    for (int i = 0; i < n; ++i) {
      if (v[i].model_id != model_id || v[i].nth != nth || v[i].ne != ne || v[i].nc < nc_needed) continue;
      if (v[i].lds_bytes(N) > 160 * 1024) continue;
      if (!best || v[i].big < best->big || (v[i].big == best->big && v[i].nc < best->nc)) best = &v[i];
    }
  }
  return best;
}

}  // namespace hilo
