// Tracking NMPC for horizons whose iterate does not fit the 160 KB of LDS: same policy and engine, the iterate lives in a
// per-instance workspace in global memory (Ocp<PB>::BIG; hilo_ocp.h).
#include "hilo_nmpc_gen.h"
#include "hilo_nmpc_track.h"

namespace hilo {

const TrackBigVariant* nmpc_track_big_find(int model_id) {
#define V(ID, M) {ID, &gen_lds<NmpcTrack<M, true, false>>, &gen_ws<NmpcTrack<M, true, false>>, &gen_launch<NmpcTrack<M, true, false>>}
  static const TrackBigVariant v[] = {
      V(HILO_MODEL_CHEMOSTAT4, Chemostat4), V(HILO_MODEL_PENDULUM4, Pendulum4), V(HILO_MODEL_BIOREACTOR3, Bioreactor3),
      V(HILO_MODEL_ROBOT6, Robot6),
  };
#undef V
  for (const auto& c : v)
    if (c.model_id == model_id) return &c;
  return nullptr;
}

}  // namespace hilo
