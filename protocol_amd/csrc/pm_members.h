// pm_members.h — the member list of a host-side group record.
//
// A full-swarm match absorbs thousands of new groups into the host list in one go (absorb_groups) and a cold match
// drops as many (pm_reset_groups); groups are small (the reference's topologies ask for 1..8 nodes, rarely more), so a
// std::vector per group spends that time in malloc / free — 40 of the ~60 us the host list costs per 2,000 groups.
// Up to kInline members live inside the record; longer lists go to the heap.  Only what pm_engine.cpp uses.
#ifndef PM_MEMBERS_H
#define PM_MEMBERS_H

#include <stddef.h>
#include <stdint.h>

#include <cstring>
#include <vector>

namespace pm {

class MemberList {
 public:
  static constexpr uint32_t kInline = 8;

  MemberList() noexcept {}
  ~MemberList() { release(); }
  MemberList(const MemberList& o) { assign(o.begin(), o.end()); }
  MemberList(MemberList&& o) noexcept { steal(o); }
  MemberList& operator=(const MemberList& o) {
    if (this != &o) assign(o.begin(), o.end());
    return *this;
  }
  MemberList& operator=(MemberList&& o) noexcept {
    if (this != &o) {
      release();
      steal(o);
    }
    return *this;
  }
  MemberList& operator=(const std::vector<uint32_t>& v) {
    assign(v.data(), v.data() + v.size());
    return *this;
  }
  operator std::vector<uint32_t>() const { return std::vector<uint32_t>(begin(), end()); }

  // [first, last) must not point into this list
  void assign(const uint32_t* first, const uint32_t* last) {
    const size_t n = size_t(last - first);
    if (n > cap_) {
      uint32_t* h = new uint32_t[n];
      release();
      heap_ = h;
      cap_ = uint32_t(n);
    }
    if (n) std::memcpy(data(), first, n * sizeof(uint32_t));
    n_ = uint32_t(n);
  }

  size_t size() const noexcept { return n_; }
  bool empty() const noexcept { return n_ == 0; }
  uint32_t* data() noexcept { return cap_ > kInline ? heap_ : in_; }
  const uint32_t* data() const noexcept { return cap_ > kInline ? heap_ : in_; }
  uint32_t* begin() noexcept { return data(); }
  uint32_t* end() noexcept { return data() + n_; }
  const uint32_t* begin() const noexcept { return data(); }
  const uint32_t* end() const noexcept { return data() + n_; }
  uint32_t operator[](size_t i) const noexcept { return data()[i]; }
  uint32_t& operator[](size_t i) noexcept { return data()[i]; }

 private:
  void release() noexcept {
    if (cap_ > kInline) delete[] heap_;
    cap_ = kInline;
    n_ = 0;
  }
  void steal(MemberList& o) noexcept {
    n_ = o.n_;
    cap_ = o.cap_;
    if (o.cap_ > kInline)
      heap_ = o.heap_;
    else
      std::memcpy(in_, o.in_, sizeof(in_));
    o.cap_ = kInline;
    o.n_ = 0;
  }

  uint32_t n_ = 0, cap_ = kInline;  // cap_ > kInline <=> the members are on the heap
  union {
    uint32_t in_[kInline];
    uint32_t* heap_;
  };
};

}  // namespace pm
#endif
