// Host interface of the register-resident equalisation engine (dfq_le_resident.hip), used by the plan of dfq_le.hip.
#pragma once

#include <string>

#include "dfq_common.hpp"
#include "dfq_le_shared.hpp"

namespace dfq {

struct LeResident;   // opaque

// nullptr (with a reason in *why_not) when the network cannot be kept resident: the caller uses the streaming kernel
LeResident* le_resident_create(const dfq_layer* layers, int n_layers, const dfq_relation* relations, int n_relations,
                               std::string* why_not);
void le_resident_destroy(LeResident* r);
int le_resident_tiles(const LeResident* r);
int le_resident_stats(const LeResident* r, hipStream_t st, int64_t* out5);   // rollbacks of the last launch (tests, tuning)
int le_resident_stored_tiles(const LeResident* r, hipStream_t st, int64_t* out);   // tiles of the last launch that stored (0: network untouched)
int64_t le_resident_elements(const LeResident* r);
// ONE launch: load, run up to n_sweeps sweeps of the loop whose state is *d_state (stops early when the reference's
// exit test fires), store.  Asynchronous on `st`.
// `restart`: the launch in front of the cooperative one also resets the loop state and the error word (dfq.py:81-82)
int le_resident_enqueue(LeResident* r, const dfq_le_config* cfg, LeState* d_state, unsigned long long* d_err, int n_sweeps,
                        hipStream_t st, long long* d_trace = nullptr, int restart = 0);
int le_resident_trace_words(const LeResident* r);   // tuning aid: int64 words of the per-tile phase stamps

}  // namespace dfq
