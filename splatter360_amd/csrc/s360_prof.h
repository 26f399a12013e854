// s360_prof.h — optional per-kernel timing with HIP events recorded on the launch stream itself
// (bench.py's roofline figures come from here).  Disabled by default: zero cost on the hot path.
#pragma once
#include <hip/hip_runtime.h>

namespace s360 {

enum ProfSlot {
    PS_PREPROCESS = 0,
    PS_TILE_SCAN,
    PS_EMIT,
    PS_SORT,
    PS_RENDER,
    PS_RENDER_BWD,
    PS_PREPROCESS_BWD,
    PS_STITCH,
    PS_STITCH_BWD,
    PS_SH_EVAL,
    PS_GATHER,
    PS_SH_BWD,
    PS_ORDER,
    PS_NSLOTS
};

bool prof_enabled();
void prof_mark(int slot, hipStream_t st, bool end);

struct ProfScope {
    int slot;
    hipStream_t st;
    bool on;
    ProfScope(int s, hipStream_t stream) : slot(s), st(stream), on(prof_enabled()) {
        if (on) prof_mark(slot, st, false);
    }
    ~ProfScope() {
        if (on) prof_mark(slot, st, true);
    }
};

}  // namespace s360
