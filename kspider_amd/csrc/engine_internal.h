// Internal declarations shared by the engine (engine.hip) and the host-side
// reference surface (pairwise_host.cpp).  Not part of the C ABI.
#ifndef KSPIDER_ENGINE_INTERNAL_H
#define KSPIDER_ENGINE_INTERNAL_H
#include <cstdint>
#include <string>

#include "../../include/kspider_amd.h"

#include <vector>

namespace ksp {
void set_error(const std::string& s);
// GPUs of a drop-in call: $KSPIDER_DEVICES ("0,1,2,..."; a device may appear twice) or the one of $KSPIDER_DEVICE (0)
std::vector<int> devices_from_env();
}

extern "C" {
/* test/diagnostic hook: distinct-key offsets of the block lists (nb + 1 values). */
int ksp_engine_block_key_counts(const ksp_engine* e, uint32_t* h_blk_off);
int ksp_engine_source_order(const ksp_engine* e, uint32_t* h_newidx);   // (diagnostics) engine index of every source
}
#endif
