#include "common.h"
#include "engine.h"
namespace d4 {
int learn(d4_engine* e, const d4_learn_io* io, hipStream_t s) { D4_REQUIRE(false, "learn: not built yet"); }
int optim_step(d4_engine*, int, float*, int, float, float, float, float, float, float, float, float*, hipStream_t) { D4_REQUIRE(false, "optim: not built yet"); }
int64_t group_numel(const d4_engine*, int) { return 0; }
}
extern "C" int d4_gae(const float*, const float*, const int64_t*, const uint8_t*, const uint8_t*, float, float, int, int, float*, void*) { return 9; }
