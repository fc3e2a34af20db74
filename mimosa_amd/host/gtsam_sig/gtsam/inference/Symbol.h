// gtsam_sig: stand-in for <gtsam/inference/Symbol.h>: the character-tagged keys (symbol_shorthand).  NOT GTSAM.
#pragma once
#include <gtsam/inference/Key.h>

namespace gtsam
{
inline Key symbol(unsigned char c, std::uint64_t j) { return (static_cast<Key>(c) << 56) | j; }
namespace symbol_shorthand
{
inline Key X(std::uint64_t j) { return symbol('x', j); }
inline Key G(std::uint64_t j) { return symbol('g', j); }
inline Key V(std::uint64_t j) { return symbol('v', j); }
inline Key B(std::uint64_t j) { return symbol('b', j); }
}  // namespace symbol_shorthand
}  // namespace gtsam
