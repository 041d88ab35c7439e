// pm_host.cpp — host-side mirror of the reference's configuration front end (no GPU needed):
//   ComputeRequirements::from_str      crates/shared/src/models/node.rs:180-374
//   GpuSpecs::meets model-string rule  crates/shared/src/models/node.rs:463-484
//   config priority order              crates/orchestrator/src/plugins/node_groups/mod.rs:150-164, :399-418
// This is product code: it shares nothing with oracle/, which restates the same rules separately so
// the two can be checked against each other and against the reference's known-answer tests.
#include <algorithm>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "pm_engine.h"
#include "pm_internal.h"

namespace pm {

namespace {

#include "pm_unicode_lower.inc"  // GENERATED (tools/make_unicode_tables.py): pm_uc_lower_map, pm_uc_cased_ranges, pm_uc_case_ignorable_ranges

// One code point of a UTF-8 string (Rust strings are valid UTF-8; a byte that is not the start of a well-formed sequence is
// passed through as itself: *len = 1, value = 0x110000 + byte, which no table knows).
uint32_t decode_utf8(std::string_view s, size_t i, size_t* len) {
  const auto b = [&](size_t k) { return uint32_t(static_cast<unsigned char>(s[k])); };
  const uint32_t c = b(i);
  *len = 1;
  if (c < 0x80u) return c;
  const auto cont = [&](size_t k) { return k < s.size() && (b(k) & 0xC0u) == 0x80u; };
  if (c >= 0xC2u && c <= 0xDFu && cont(i + 1)) {
    *len = 2;
    return ((c & 0x1Fu) << 6) | (b(i + 1) & 0x3Fu);
  }
  if (c >= 0xE0u && c <= 0xEFu && cont(i + 1) && cont(i + 2)) {
    const uint32_t v = ((c & 0x0Fu) << 12) | ((b(i + 1) & 0x3Fu) << 6) | (b(i + 2) & 0x3Fu);
    if (v >= 0x800u && !(v >= 0xD800u && v <= 0xDFFFu)) {
      *len = 3;
      return v;
    }
  }
  if (c >= 0xF0u && c <= 0xF4u && cont(i + 1) && cont(i + 2) && cont(i + 3)) {
    const uint32_t v = ((c & 0x07u) << 18) | ((b(i + 1) & 0x3Fu) << 12) | ((b(i + 2) & 0x3Fu) << 6) | (b(i + 3) & 0x3Fu);
    if (v >= 0x10000u && v <= 0x10FFFFu) {
      *len = 4;
      return v;
    }
  }
  return 0x110000u + c;
}
void encode_utf8(uint32_t v, std::string* o) {
  if (v < 0x80u) o->push_back(char(v));
  else if (v < 0x800u) { o->push_back(char(0xC0u | (v >> 6))); o->push_back(char(0x80u | (v & 0x3Fu))); }
  else if (v < 0x10000u) { o->push_back(char(0xE0u | (v >> 12))); o->push_back(char(0x80u | ((v >> 6) & 0x3Fu))); o->push_back(char(0x80u | (v & 0x3Fu))); }
  else { o->push_back(char(0xF0u | (v >> 18))); o->push_back(char(0x80u | ((v >> 12) & 0x3Fu))); o->push_back(char(0x80u | ((v >> 6) & 0x3Fu))); o->push_back(char(0x80u | (v & 0x3Fu))); }
}

// char::is_whitespace (the White_Space property): what str::trim strips
bool is_whitespace(uint32_t c) {
  return (c >= 0x09u && c <= 0x0Du) || c == 0x20u || c == 0x85u || c == 0xA0u || c == 0x1680u || (c >= 0x2000u && c <= 0x200Au) ||
         c == 0x2028u || c == 0x2029u || c == 0x202Fu || c == 0x205Fu || c == 0x3000u;
}

template <size_t N>
bool in_ranges(const uint32_t (&r)[N][2], uint32_t c) {
  size_t lo = 0, hi = N;
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (c > r[mid][1]) lo = mid + 1;
    else hi = mid;
  }
  return lo < N && c >= r[lo][0];
}

std::string_view trim(std::string_view s) {  // str::trim
  size_t b = 0, e = s.size();
  while (b < e) {
    size_t n;
    if (!is_whitespace(decode_utf8(s, b, &n))) break;
    b += n;
  }
  while (e > b) {  // (the last code point: step back over continuation bytes)
    size_t k = e - 1;
    while (k > b && (static_cast<unsigned char>(s[k]) & 0xC0u) == 0x80u && e - k < 4) --k;
    size_t n;
    const uint32_t c = decode_utf8(s, k, &n);
    if (k + n != e || !is_whitespace(c)) break;
    e = k;
  }
  return s.substr(b, e - b);
}

// alloc::str: case_ignorable_then_cased over the code points of s[from..to) (forwards) or backwards from `from`
bool cased_after_ignorables_fwd(std::string_view s, size_t i) {
  while (i < s.size()) {
    size_t n;
    const uint32_t c = decode_utf8(s, i, &n);
    if (!in_ranges(pm_uc_case_ignorable_ranges, c)) return in_ranges(pm_uc_cased_ranges, c);
    i += n;
  }
  return false;
}
bool cased_after_ignorables_back(std::string_view s, size_t i) {
  while (i > 0) {
    size_t k = i - 1;
    while (k > 0 && (static_cast<unsigned char>(s[k]) & 0xC0u) == 0x80u && i - k < 4) --k;
    size_t n;
    uint32_t c = decode_utf8(s, k, &n);
    if (k + n != i) {  // (a stray continuation byte: a code point of its own)
      k = i - 1;
      c = decode_utf8(s, k, &n);
    }
    if (!in_ranges(pm_uc_case_ignorable_ranges, c)) return in_ranges(pm_uc_cased_ranges, c);
    i = k;
  }
  return false;
}

}  // namespace

// str::to_lowercase (node.rs:465, :470): every code point through char::to_lowercase — which may give up to three — and
// a capital sigma at the end of a word as the final form (alloc::str::to_lowercase, map_uppercase_sigma)
std::string to_lowercase(std::string_view s) {
  std::string o;
  o.reserve(s.size());
  for (size_t i = 0; i < s.size();) {
    size_t n;
    const uint32_t c = decode_utf8(s, i, &n);
    if (c < 0x80u) {
      o.push_back(c >= 'A' && c <= 'Z' ? char(c - 'A' + 'a') : char(c));
    } else if (c == 0x3A3u) {
      const bool word_final = cased_after_ignorables_back(s, i) && !cased_after_ignorables_fwd(s, i + n);
      encode_utf8(word_final ? 0x3C2u : 0x3C3u, &o);
    } else {
      constexpr size_t N = sizeof(pm_uc_lower_map) / sizeof(pm_uc_lower_map[0]);
      size_t lo = 0, hi = N;
      while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (pm_uc_lower_map[mid].cp < c) lo = mid + 1;
        else hi = mid;
      }
      if (lo < N && pm_uc_lower_map[lo].cp == c)
        for (uint32_t k = 0; k < pm_uc_lower_map[lo].n; ++k) encode_utf8(pm_uc_lower_map[lo].out[k], &o);
      else
        o.append(s.substr(i, n));
    }
    i += n;
  }
  return o;
}

namespace {

// <u32 as FromStr>: optional '+', at least one ASCII digit, no overflow.
bool parse_u32(std::string_view s, uint32_t* out) {
  if (!s.empty() && s.front() == '+') s.remove_prefix(1);
  if (s.empty()) return false;
  uint64_t v = 0;
  for (char c : s) {
    if (c < '0' || c > '9') return false;
    v = v * 10 + uint64_t(c - '0');
    if (v > 0xFFFFFFFFull) return false;
  }
  *out = uint32_t(v);
  return true;
}

std::string normalize_model(std::string_view s) {  // to_lowercase().replace(' ', "_")
  std::string o = to_lowercase(s);
  for (char& c : o)
    if (c == ' ') c = '_';
  return o;
}
std::string drop_underscores(const std::string& s) {
  std::string o;
  o.reserve(s.size());
  for (char c : s)
    if (c != '_') o.push_back(c);
  return o;
}
bool contains(const std::string& hay, const std::string& needle) { return hay.find(needle) != std::string::npos; }

}  // namespace

bool model_matches(std::string_view spec_model, std::string_view req_model) {
  const std::string ns = normalize_model(spec_model);  // node.rs:465
  const std::string ns_nu = drop_underscores(ns);      // :473
  size_t pos = 0;
  for (;;) {  // req_model.split(',') — an empty string still yields one (empty) part
    const size_t comma = req_model.find(',', pos);
    const std::string_view part =
        req_model.substr(pos, comma == std::string_view::npos ? std::string_view::npos : comma - pos);
    const std::string nr = normalize_model(trim(part));  // :468-470
    const std::string nr_nu = drop_underscores(nr);      // :474
    if (contains(ns, nr) || contains(nr, ns) || contains(ns_nu, nr_nu) || contains(nr_nu, ns_nu)) return true;
    if (comma == std::string_view::npos) break;
    pos = comma + 1;
  }
  return false;
}

// Parsed requirement with owned model strings.
int32_t parse_requirements(std::string_view s, ParsedRequirements* out) {
  out->flags = 0;
  out->cpu_cores = out->ram_mb = out->storage_gb = 0;
  out->alts.clear();
  out->models.clear();
  pm_gpu_alt_row cur{};
  std::string cur_model;
  bool started = false;  // gpu_spec_started, node.rs:185
  auto push_cur = [&]() {
    out->alts.push_back(cur);
    out->models.push_back(cur_model);
    cur = pm_gpu_alt_row{};
    cur_model.clear();
  };
  size_t pos = 0;
  for (;;) {
    const size_t semi = s.find(';', pos);
    std::string_view part = trim(s.substr(pos, semi == std::string_view::npos ? std::string_view::npos : semi - pos));
    if (!part.empty()) {
      const size_t eq = part.find('=');  // splitn(2, '=')
      if (eq == std::string_view::npos) return set_error(PM_EPARSE, "Invalid key-value pair format");
      const std::string_view key = trim(part.substr(0, eq));
      const std::string_view value = trim(part.substr(eq + 1));
      uint32_t v = 0;
      const bool num = parse_u32(value, &v);
      if (key == "gpu:count") {  // :203-216
        if (started && (cur.flags & PM_G_COUNT)) push_cur();
        started = true;
        if (!num) return set_error(PM_EPARSE, "Invalid gpu:count value");
        cur.flags |= PM_G_COUNT;
        cur.count = v;
      } else if (key == "gpu:model") {  // :217-222
        started = true;
        cur.flags |= PM_G_MODEL;
        cur_model.assign(value);
      } else if (key == "gpu:memory_mb") {  // :223-238
        started = true;
        if (cur.flags & (PM_G_MEM_MIN | PM_G_MEM_MAX))
          return set_error(PM_EPARSE, "Cannot specify both exact memory and min/max memory");
        if (!num) return set_error(PM_EPARSE, "Invalid gpu:memory_mb value");
        cur.flags |= PM_G_MEM;
        cur.memory_mb = v;
      } else if (key == "gpu:memory_mb_min") {  // :239-262
        started = true;
        if (cur.flags & PM_G_MEM) return set_error(PM_EPARSE, "Cannot specify both exact memory and min/max memory");
        if (cur.flags & PM_G_MEM_MAX) {
          if (!num) return set_error(PM_EPANIC, "reference panics: unwrap on non-numeric gpu:memory_mb_min");
          if (cur.memory_mb_max < v) return set_error(PM_EPARSE, "min value is greater than max value");
        }
        if (!num) return set_error(PM_EPARSE, "Invalid gpu:memory_mb_min value");
        cur.flags |= PM_G_MEM_MIN;
        cur.memory_mb_min = v;
      } else if (key == "gpu:memory_mb_max") {  // :263-288
        started = true;
        if (cur.flags & PM_G_MEM) return set_error(PM_EPARSE, "Cannot specify both exact memory and min/max memory");
        if (cur.flags & PM_G_MEM_MIN) {
          if (!num) return set_error(PM_EPANIC, "reference panics: unwrap on non-numeric gpu:memory_mb_max");
          if (cur.memory_mb_min > v) return set_error(PM_EPARSE, "max value is less than min value");
        }
        if (!num) return set_error(PM_EPARSE, "Invalid gpu:memory_mb_max value");
        cur.flags |= PM_G_MEM_MAX;
        cur.memory_mb_max = v;
      } else if (key == "gpu:total_memory_min") {  // :290-310
        started = true;
        if (cur.flags & PM_G_TOT_MAX) {
          if (!num) return set_error(PM_EPANIC, "reference panics: unwrap on non-numeric gpu:total_memory_min");
          if (cur.total_memory_max < v) return set_error(PM_EPARSE, "min value is greater than max value");
        }
        if (!num) return set_error(PM_EPARSE, "Invalid gpu:total_memory_min value");
        cur.flags |= PM_G_TOT_MIN;
        cur.total_memory_min = v;
      } else if (key == "gpu:total_memory_max") {  // :311-331
        started = true;
        if (cur.flags & PM_G_TOT_MIN) {
          if (!num) return set_error(PM_EPANIC, "reference panics: unwrap on non-numeric gpu:total_memory_max");
          if (cur.total_memory_min > v) return set_error(PM_EPARSE, "max value is less than min value");
        }
        if (!num) return set_error(PM_EPARSE, "Invalid gpu:total_memory_max value");
        cur.flags |= PM_G_TOT_MAX;
        cur.total_memory_max = v;
      } else if (key == "cpu:cores") {  // :333-341
        if (!num) return set_error(PM_EPARSE, "Invalid cpu:cores value");
        out->flags |= PM_R_CPU | PM_R_CPU_CORES;
        out->cpu_cores = v;
      } else if (key == "ram_mb") {  // :344-350
        if (!num) return set_error(PM_EPARSE, "Invalid ram_mb value");
        out->flags |= PM_R_RAM;
        out->ram_mb = v;
      } else if (key == "storage_gb") {  // :351-357
        if (!num) return set_error(PM_EPARSE, "Invalid storage_gb value");
        out->flags |= PM_R_STORAGE;
        out->storage_gb = v;
      } else {
        return set_error(PM_EPARSE, "Unknown requirement key");  // :358
      }
    }
    if (semi == std::string_view::npos) break;
    pos = semi + 1;
  }
  if (started && cur.flags != 0) push_cur();  // :360-370
  return PM_OK;
}

// Constructor sort (mod.rs:150-164): min_group_size desc, then with-requirements first; stable.
void template_order(const pm_config_row* cfgs, uint32_t n, std::vector<uint32_t>* order) {
  order->resize(n);
  for (uint32_t i = 0; i < n; ++i) (*order)[i] = i;
  std::stable_sort(order->begin(), order->end(), [&](uint32_t a, uint32_t b) {
    if (cfgs[a].min_group_size != cfgs[b].min_group_size) return cfgs[a].min_group_size > cfgs[b].min_group_size;
    return (cfgs[a].flags & PM_R_HAS_REQ) && !(cfgs[b].flags & PM_R_HAS_REQ);
  });
}

// get_available_configurations (mod.rs:399-418): enabled filter, then stable min_group_size desc.
void available_order(const pm_config_row* cfgs, uint32_t n, uint64_t enabled, std::vector<uint32_t>* out) {
  std::vector<uint32_t> tmpl;
  template_order(cfgs, n, &tmpl);
  out->clear();
  for (uint32_t c : tmpl)
    if ((enabled >> c) & 1ull) out->push_back(c);
  std::stable_sort(out->begin(), out->end(),
                   [&](uint32_t a, uint32_t b) { return cfgs[a].min_group_size > cfgs[b].min_group_size; });
}

}  // namespace pm

extern "C" {

int32_t pm_host_parse_requirements(const char* s, pm_config_row* cfg, pm_gpu_alt_row* alts, uint32_t alt_cap,
                                   char* models_out, size_t models_cap) {
  if (!s || !cfg) return pm::set_error(PM_EINVAL, "null argument");
  pm::ParsedRequirements pr;
  const int32_t rc = pm::parse_requirements(s, &pr);
  if (rc != PM_OK) return rc;
  if (pr.alts.size() > alt_cap) return pm::set_error(PM_ERANGE, "too many GPU alternatives for the caller buffer");
  cfg->flags = (cfg->flags & ~uint32_t(PM_R_CPU | PM_R_CPU_CORES | PM_R_RAM | PM_R_STORAGE)) | pr.flags | PM_R_HAS_REQ;
  cfg->cpu_cores = pr.cpu_cores;
  cfg->ram_mb = pr.ram_mb;
  cfg->storage_gb = pr.storage_gb;
  cfg->alt_count = uint32_t(pr.alts.size());
  size_t off = 0;
  for (size_t i = 0; i < pr.alts.size(); ++i) {
    pm_gpu_alt_row a = pr.alts[i];
    a.model_row = 0;
    if (a.flags & PM_G_MODEL) {
      const std::string& m = pr.models[i];
      if (!models_out || off + m.size() + 1 > models_cap)
        return pm::set_error(PM_ERANGE, "model string buffer too small");
      std::memcpy(models_out + off, m.c_str(), m.size() + 1);
      a.model_row = uint32_t(off);
      off += m.size() + 1;
    }
    if (alts) alts[i] = a;
  }
  return PM_OK;
}

int32_t pm_host_model_matches(const char* spec_model, const char* req_model) {
  if (!spec_model || !req_model) return pm::set_error(PM_EINVAL, "null argument");
  return pm::model_matches(spec_model, req_model) ? 1 : 0;
}

int32_t pm_host_to_lowercase(const char* in, char* out, size_t cap, size_t* needed) {
  if (!in || !needed) return pm::set_error(PM_EINVAL, "null argument");
  const std::string r = pm::to_lowercase(in);
  *needed = r.size() + 1;
  if (!out && cap == 0) return PM_OK;  // sizing call
  if (!out || cap < r.size() + 1) return pm::set_error(PM_ERANGE, "output buffer too small");
  std::memcpy(out, r.c_str(), r.size() + 1);
  return PM_OK;
}

int32_t pm_host_build_model_table(const char* const* req_models, uint32_t n_rows, const char* const* spec_models,
                                  uint32_t n_classes, uint32_t* bits_out) {
  if ((n_rows && !req_models) || (n_classes && !spec_models) || !bits_out)
    return pm::set_error(PM_EINVAL, "null argument");
  const uint32_t words = (n_classes + 31u) / 32u;
  std::memset(bits_out, 0, sizeof(uint32_t) * size_t(n_rows) * words);
  for (uint32_t r = 0; r < n_rows; ++r)
    for (uint32_t c = 0; c < n_classes; ++c)
      if (pm::model_matches(spec_models[c], req_models[r])) bits_out[size_t(r) * words + (c >> 5)] |= 1u << (c & 31u);
  return PM_OK;
}

int32_t pm_host_config_order(const pm_config_row* cfgs, uint32_t n_cfgs, uint64_t enabled, uint32_t* order_out,
                             uint32_t* n_out) {
  if ((n_cfgs && !cfgs) || !order_out || !n_out) return pm::set_error(PM_EINVAL, "null argument");
  if (n_cfgs > PM_MAX_CONFIGS) return pm::set_error(PM_EINVAL, "more than PM_MAX_CONFIGS configurations");
  for (uint32_t i = 0; i < n_cfgs; ++i)
    if (cfgs[i].max_group_size < cfgs[i].min_group_size)
      return pm::set_error(PM_EINVAL, "Plugin configuration is invalid (max_group_size < min_group_size)");
  std::vector<uint32_t> o;
  pm::available_order(cfgs, n_cfgs, enabled, &o);
  for (size_t i = 0; i < o.size(); ++i) order_out[i] = o[i];
  *n_out = uint32_t(o.size());
  return PM_OK;
}

// ---- group variables (scheduler_impl.rs:155-200, storage.rs:150-215)

namespace {
// str::replace: every non-overlapping occurrence, scanning left to right; an empty pattern never occurs here
void replace_all(std::string& s, std::string_view pat, std::string_view with) {
  if (pat.empty()) return;
  std::string out;
  size_t pos = 0;
  for (;;) {
    const size_t hit = s.find(pat, pos);
    if (hit == std::string::npos) break;
    out.append(s, pos, hit - pos);
    out.append(with);
    pos = hit + pat.size();
  }
  if (pos == 0) return;  // no occurrence
  out.append(s, pos, std::string::npos);
  s.swap(out);
}
int32_t give(const std::string& r, char* out, size_t cap, size_t* needed) {
  if (needed) *needed = r.size() + 1;
  if (cap < r.size() + 1 || !out) {
    if (!out && cap == 0 && needed) return PM_OK;  // sizing call
    return pm::set_error(PM_ERANGE, "output buffer too small");
  }
  std::memcpy(out, r.c_str(), r.size() + 1);
  return PM_OK;
}
}  // namespace

uint32_t pm_host_last_file_idx(const char* total_upload_count) {
  // u32::from_str: optional '+', at least one ASCII digit, nothing else, no overflow; Err => 0
  if (!total_upload_count) return 0;
  const char* p = total_upload_count;
  if (*p == '+') ++p;
  if (!*p) return 0;
  uint64_t v = 0;
  for (; *p; ++p) {
    if (*p < '0' || *p > '9') return 0;
    v = v * 10 + uint64_t(*p - '0');
    if (v > 0xFFFFFFFFull) return 0;
  }
  return v ? uint32_t(v - 1) : 0u;  // saturating_sub(1)
}

int32_t pm_host_group_vars(const char* in, const pm_group_vars* v, char* out, size_t cap, size_t* needed) {
  if (!in || !v || !v->next_p2p_address || !v->group_id || !v->total_upload_count)
    return pm::set_error(PM_EINVAL, "null argument");
  std::string s(in);
  replace_all(s, "${GROUP_INDEX}", std::to_string(v->group_index));
  replace_all(s, "${GROUP_SIZE}", std::to_string(v->group_size));
  replace_all(s, "${NEXT_P2P_ADDRESS}", v->next_p2p_address);
  replace_all(s, "${GROUP_ID}", v->group_id);
  replace_all(s, "${TOTAL_UPLOAD_COUNT}", v->total_upload_count);
  replace_all(s, "${LAST_FILE_IDX}", std::to_string(pm_host_last_file_idx(v->total_upload_count)));
  return give(s, out, cap, needed);
}

int32_t pm_host_volume_vars(const char* in, const char* group_id, char* out, size_t cap, size_t* needed) {
  if (!in || !group_id) return pm::set_error(PM_EINVAL, "null argument");
  std::string s(in);
  replace_all(s, "${GROUP_ID}", group_id);
  return give(s, out, cap, needed);
}

int32_t pm_host_upload_name_vars(const char* in, const char* group_id, uint32_t group_size, uint32_t group_index,
                                 uint64_t upload_count, char* out, size_t cap, size_t* needed) {
  if (!in) return pm::set_error(PM_EINVAL, "null argument");
  std::string s(in);
  if (group_id) {  // storage.rs:150-158: only for a node that is in a group
    replace_all(s, "${NODE_GROUP_ID}", group_id);
    replace_all(s, "${NODE_GROUP_SIZE}", std::to_string(group_size));
    replace_all(s, "${NODE_GROUP_INDEX}", std::to_string(group_index));
  }
  replace_all(s, "${TOTAL_UPLOAD_COUNT_AFTER}", std::to_string(upload_count));
  replace_all(s, "${CURRENT_FILE_INDEX}", std::to_string(upload_count ? upload_count - 1 : 0));
  return give(s, out, cap, needed);
}

uint32_t pm_abi_version(void) { return PM_ABI_VERSION; }

}  // extern "C"
