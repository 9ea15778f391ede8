// Host-side helpers shared by the translation units of libsupnerf_hip.so.
#pragma once
#include "../../include/supnerf_hip.h"
namespace snr { struct RayGeom; }
// validate snr_render_args and decode it into the by-value kernel argument
int snr_fill_geom_(const snr_render_args* a, snr::RayGeom* g, int need_model);
extern "C" int snr_check_launch_(void);
