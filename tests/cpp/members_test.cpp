// CPU unit test of pm::MemberList (protocol_amd/csrc/pm_members.h) against std::vector as the model: random
// assignments, copies, moves and growth of a std::vector<Group-like record>, as absorb_groups / compact_groups /
// run_merge use it.  Built and run by tests/test_host_helpers.py (g++, with the address sanitizer when it links).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include "pm_members.h"

struct Rec {
  uint64_t id;
  pm::MemberList members;
  bool dead = false;
};
static_assert(std::is_nothrow_move_constructible<Rec>::value, "std::vector<Group> must move, not copy, when it grows");

static uint64_t rng_state = 0x1234567ull;
static uint32_t rnd(uint32_t n) {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return uint32_t(rng_state >> 33) % n;
}
static std::vector<uint32_t> random_list() {
  static const uint32_t sizes[] = {0, 1, 2, 7, 8, 9, 16, 63, 300};
  std::vector<uint32_t> v(sizes[rnd(9)]);
  for (uint32_t& x : v) x = rnd(1u << 30);
  return v;
}
#define CHECK(c)                                              \
  do {                                                        \
    if (!(c)) {                                               \
      std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); \
      return 1;                                               \
    }                                                         \
  } while (0)
static bool same(const pm::MemberList& a, const std::vector<uint32_t>& b) {
  return a.size() == b.size() && std::equal(a.begin(), a.end(), b.begin()) && a.empty() == b.empty();
}

int main() {
  std::vector<Rec> recs;
  std::vector<std::vector<uint32_t>> model;
  for (int step = 0; step < 200000; ++step) {
    const uint32_t op = rnd(10);
    if (op < 3 || recs.empty()) {  // absorb_groups: a new record, moved into the list
      std::vector<uint32_t> v = random_list();
      Rec r;
      r.id = step;
      r.members.assign(v.data(), v.data() + v.size());
      for (uint32_t w : r.members) (void)w;
      recs.push_back(std::move(r));
      model.push_back(v);
    } else if (op == 3) {  // run_merge: members = a vector
      const uint32_t i = rnd(uint32_t(recs.size()));
      std::vector<uint32_t> v = random_list();
      recs[i].members = v;
      model[i] = v;
    } else if (op == 4) {  // copies (pm_on_worker_status keeps a copy of a dissolved group's members)
      const uint32_t i = rnd(uint32_t(recs.size())), j = rnd(uint32_t(recs.size()));
      const std::vector<uint32_t> as_vec = recs[i].members;
      CHECK(as_vec == model[i]);
      pm::MemberList c(recs[i].members);
      CHECK(same(c, model[i]));
      recs[j].members = c;  // (i == j: self-assignment through a copy)
      model[j] = model[i];
      recs[j].members = recs[j].members;
      CHECK(same(recs[j].members, model[j]));
    } else if (op == 5) {  // moves
      const uint32_t i = rnd(uint32_t(recs.size())), j = rnd(uint32_t(recs.size()));
      if (i != j) {
        recs[j].members = std::move(recs[i].members);
        model[j] = model[i];
        model[i].clear();
        CHECK(recs[i].members.size() == 0);
      }
    } else if (op == 6 && rnd(60) == 0) {  // compact_groups: remove_if + erase
      for (size_t k = 0; k < recs.size(); ++k)
        if (rnd(4) == 0) recs[k].dead = true;
      std::vector<std::vector<uint32_t>> kept;
      for (size_t k = 0; k < recs.size(); ++k)
        if (!recs[k].dead) kept.push_back(model[k]);
      recs.erase(std::remove_if(recs.begin(), recs.end(), [](const Rec& r) { return r.dead; }), recs.end());
      model.swap(kept);
    } else if (op == 7 && recs.size() > 3000) {  // pm_reset_groups
      recs.clear();
      model.clear();
    } else {  // element access and the iteration pm_get_groups / push_groups do
      const uint32_t i = rnd(uint32_t(recs.size()));
      std::vector<uint32_t> flat;
      flat.insert(flat.end(), recs[i].members.begin(), recs[i].members.end());
      CHECK(flat == model[i]);
      if (!model[i].empty()) CHECK(recs[i].members[0] == model[i][0]);
      std::vector<uint32_t> sorted_m = recs[i].members;
      std::sort(sorted_m.begin(), sorted_m.end());
    }
    if (step % 997 == 0) {
      CHECK(recs.size() == model.size());
      for (size_t k = 0; k < recs.size(); ++k) CHECK(same(recs[k].members, model[k]));
    }
  }
  CHECK(recs.size() == model.size());
  for (size_t k = 0; k < recs.size(); ++k) CHECK(same(recs[k].members, model[k]));
  std::printf("members_test ok (%zu records at the end)\n", recs.size());
  return 0;
}
