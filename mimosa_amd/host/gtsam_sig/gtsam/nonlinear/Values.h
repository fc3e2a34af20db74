// gtsam_sig: stand-in for <gtsam/nonlinear/Values.h>: typed values by key, the members the host mirror calls.  NOT GTSAM.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <typeinfo>

#include <gtsam/inference/Key.h>

namespace gtsam
{
class Values
{
public:
  template <typename T>
  void insert(Key j, const T & v)
  {
    if (!m_.emplace(j, Slot{std::make_shared<T>(v), &typeid(T)}).second) throw std::invalid_argument("Values::insert: key already exists");
  }
  template <typename T>
  void update(Key j, const T & v)
  {
    auto it = m_.find(j);
    if (it == m_.end()) throw std::out_of_range("Values::update: key does not exist");
    it->second = Slot{std::make_shared<T>(v), &typeid(T)};
  }
  template <typename T>
  void insert_or_assign(Key j, const T & v)
  {
    m_[j] = Slot{std::make_shared<T>(v), &typeid(T)};
  }
  template <typename T>
  const T & at(Key j) const
  {
    auto it = m_.find(j);
    if (it == m_.end()) throw std::out_of_range("Values::at: key does not exist");
    if (*it->second.type != typeid(T)) throw std::invalid_argument("Values::at: wrong type for key");
    return *static_cast<const T *>(it->second.p.get());
  }
  bool exists(Key j) const { return m_.count(j) != 0; }
  size_t size() const { return m_.size(); }

private:
  struct Slot
  {
    std::shared_ptr<void> p;
    const std::type_info * type;
  };
  std::map<Key, Slot> m_;
};
}  // namespace gtsam
