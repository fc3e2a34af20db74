// gtsam_sig: stand-in for <gtsam/inference/Key.h>.  NOT GTSAM.
#pragma once
#include <cstdint>
#include <vector>

namespace gtsam
{
typedef std::uint64_t Key;
typedef std::vector<Key> KeyVector;
}  // namespace gtsam
