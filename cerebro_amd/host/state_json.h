// state_json.h -- reader for the descriptor part of cerebro's on-disk checkpoint.
//
// Format written by DataManager::saveStateToDisk (/root/reference/src/DataManager.cpp:1098-1215):
//   { "DataNodes": [ { "stampNSec": <u64>, "isWholeImageDescriptorAvailable": <bool>,
//                      "wholeImageDescriptor": { "rows": D, "cols": 1, "data": "v0\nv1\n..." }, ... }, ... ],
//     "ImageDataManager": ... }
// `data` is Eigen's IOFormat(FullPrecision, DontAlignCols, ", ", "\n") text (:1121,:1157-1168), parsed back by the
// reference with std::stod (src/utils/RawFileIO.cpp:418-459); nodes are stored in data_map (time-stamp) order and
// Cerebro rebuilds wholeImageComputedList from the nodes that have a descriptor, in that order (src/Cerebro.cpp:133-161).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace cerebro_hip {

struct StateDescriptors {
    int D = 0;                       // descriptor length (rows)
    std::vector<uint64_t> stampNSec; // one per descriptor, file order
    std::vector<double> desc;        // stampNSec.size() * D values
    int64_t n_nodes = 0;             // DataNodes seen (with or without descriptor)
    std::vector<uint64_t> all_stampNSec; // stamp of EVERY DataNode that has one, file order: the keys of DataManager's data_map
    std::string error;               // non-empty on failure
};

// Parses `state.json` text.  Returns false (and sets out.error) on malformed input or inconsistent descriptor sizes.
bool parse_state_json(const std::string &text, StateDescriptors &out);
bool load_state_json(const std::string &path, StateDescriptors &out);

}  // namespace cerebro_hip
