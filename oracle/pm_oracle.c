/*
 * pm_oracle.c — CPU restatement of the reference allocation path.  TEST INFRASTRUCTURE ONLY
 * (see pm_oracle.h for who may call this and for the parity-pinning statement).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off keeps a*b+c as two roundings like rustc does (no FMA fusion), and sin/cos/
 * atan2/sqrt come from the system glibc libm, which is what Rust's f64 methods call on Linux.
 *
 * All citations are relative to /root/reference/crates.
 */
#define _GNU_SOURCE
#include "pm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ small helpers */

uint64_t orc_splitmix64(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static uint64_t mix64(uint64_t x) {
  uint64_t s = x;
  return orc_splitmix64(&s);
}

#include "orc_unicode_lower.inc" /* GENERATED (tools/make_unicode_tables.py): orc_uc_lower_map, orc_uc_cased_ranges, orc_uc_case_ignorable_ranges */

/* One code point of UTF-8 text in [p, e); a byte that starts no well-formed sequence counts as a code point of its own
 * (0x110000 + byte): Rust strings are valid UTF-8, the C boundary does not promise it. */
static uint32_t utf8_next(const char* p, const char* e, size_t* len) {
  const unsigned char* u = (const unsigned char*)p;
  size_t left = (size_t)(e - p);
  uint32_t c = u[0];
  *len = 1;
  if (c < 0x80u) return c;
  if (c >= 0xC2u && c <= 0xDFu && left >= 2 && (u[1] & 0xC0u) == 0x80u) {
    *len = 2;
    return ((c & 0x1Fu) << 6) | (u[1] & 0x3Fu);
  }
  if (c >= 0xE0u && c <= 0xEFu && left >= 3 && (u[1] & 0xC0u) == 0x80u && (u[2] & 0xC0u) == 0x80u) {
    uint32_t v = ((c & 0x0Fu) << 12) | ((uint32_t)(u[1] & 0x3Fu) << 6) | (u[2] & 0x3Fu);
    if (v >= 0x800u && !(v >= 0xD800u && v <= 0xDFFFu)) {
      *len = 3;
      return v;
    }
  }
  if (c >= 0xF0u && c <= 0xF4u && left >= 4 && (u[1] & 0xC0u) == 0x80u && (u[2] & 0xC0u) == 0x80u && (u[3] & 0xC0u) == 0x80u) {
    uint32_t v = ((c & 0x07u) << 18) | ((uint32_t)(u[1] & 0x3Fu) << 12) | ((uint32_t)(u[2] & 0x3Fu) << 6) | (u[3] & 0x3Fu);
    if (v >= 0x10000u && v <= 0x10FFFFu) {
      *len = 4;
      return v;
    }
  }
  return 0x110000u + c;
}
/* the code point that ENDS at e (p < e): its start */
static const char* utf8_prev(const char* b, const char* e, uint32_t* cp) {
  const char* k = e - 1;
  size_t n;
  while (k > b && ((unsigned char)*k & 0xC0u) == 0x80u && e - k < 4) --k;
  *cp = utf8_next(k, e, &n);
  if (k + n != e) { /* a stray continuation byte */
    k = e - 1;
    *cp = utf8_next(k, e, &n);
  }
  return k;
}
static size_t utf8_put(uint32_t v, char* o) {
  if (v < 0x80u) { o[0] = (char)v; return 1; }
  if (v < 0x800u) { o[0] = (char)(0xC0u | (v >> 6)); o[1] = (char)(0x80u | (v & 0x3Fu)); return 2; }
  if (v < 0x10000u) { o[0] = (char)(0xE0u | (v >> 12)); o[1] = (char)(0x80u | ((v >> 6) & 0x3Fu)); o[2] = (char)(0x80u | (v & 0x3Fu)); return 3; }
  o[0] = (char)(0xF0u | (v >> 18)); o[1] = (char)(0x80u | ((v >> 12) & 0x3Fu)); o[2] = (char)(0x80u | ((v >> 6) & 0x3Fu)); o[3] = (char)(0x80u | (v & 0x3Fu));
  return 4;
}

/* Rust's char::is_whitespace (White_Space): what str::trim strips */
static int is_ws_cp(uint32_t c) {
  return (c >= 0x09u && c <= 0x0Du) || c == 0x20u || c == 0x85u || c == 0xA0u || c == 0x1680u || (c >= 0x2000u && c <= 0x200Au) ||
         c == 0x2028u || c == 0x2029u || c == 0x202Fu || c == 0x205Fu || c == 0x3000u;
}

/* trim [b,e) in place (str::trim) */
static void trim(const char** b, const char** e) {
  while (*b < *e) {
    size_t n;
    if (!is_ws_cp(utf8_next(*b, *e, &n))) break;
    *b += n;
  }
  while (*e > *b) {
    uint32_t c;
    const char* k = utf8_prev(*b, *e, &c);
    if (!is_ws_cp(c)) break;
    *e = k;
  }
}

static int in_ranges(const uint32_t (*r)[2], size_t n, uint32_t c) {
  size_t lo = 0, hi = n;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (c > r[mid][1]) lo = mid + 1;
    else hi = mid;
  }
  return lo < n && c >= r[lo][0];
}
#define ORC_CASED(c) in_ranges(orc_uc_cased_ranges, sizeof(orc_uc_cased_ranges) / sizeof(orc_uc_cased_ranges[0]), (c))
#define ORC_IGNORABLE(c) in_ranges(orc_uc_case_ignorable_ranges, sizeof(orc_uc_case_ignorable_ranges) / sizeof(orc_uc_case_ignorable_ranges[0]), (c))

/* str::to_lowercase (alloc::str): char::to_lowercase per code point; 'Σ' is 'ς' when it ends a word — preceded by a cased
 * letter and not followed by one, case-ignorable code points skipped on both sides (map_uppercase_sigma).  Writes at most
 * cap - 1 bytes + NUL (input that does not fit is cut at a code point boundary: the oracle's strings are short). */
static void orc_to_lowercase(const char* b, const char* e, char* out, size_t cap) {
  size_t n = 0;
  for (const char* p = b; p < e;) {
    size_t len;
    uint32_t c = utf8_next(p, e, &len);
    char tmp[16];
    size_t m = 0;
    if (c < 0x80u) {
      tmp[m++] = (c >= 'A' && c <= 'Z') ? (char)(c - 'A' + 'a') : (char)c;
    } else if (c == 0x3A3u) {
      int before = 0, after = 0;
      for (const char* q = p; q > b;) { /* backwards over the ignorables */
        uint32_t d;
        q = utf8_prev(b, q, &d);
        if (!ORC_IGNORABLE(d)) { before = ORC_CASED(d); break; }
      }
      for (const char* q = p + len; q < e;) {
        size_t l2;
        uint32_t d = utf8_next(q, e, &l2);
        if (!ORC_IGNORABLE(d)) { after = ORC_CASED(d); break; }
        q += l2;
      }
      m = utf8_put((before && !after) ? 0x3C2u : 0x3C3u, tmp);
    } else {
      size_t lo = 0, hi = sizeof(orc_uc_lower_map) / sizeof(orc_uc_lower_map[0]), N = hi;
      while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (orc_uc_lower_map[mid].cp < c) lo = mid + 1;
        else hi = mid;
      }
      if (lo < N && orc_uc_lower_map[lo].cp == c) {
        for (uint32_t k = 0; k < orc_uc_lower_map[lo].n; ++k) m += utf8_put(orc_uc_lower_map[lo].out[k], tmp + m);
      } else {
        memcpy(tmp, p, len);
        m = len;
      }
    }
    if (n + m + 1 > cap) break;
    memcpy(out + n, tmp, m);
    n += m;
    p += len;
  }
  out[n] = 0;
}

/* Rust <u32 as FromStr>::from_str: optional '+', >=1 ASCII digit, no overflow. */
static int parse_u32(const char* b, const char* e, uint32_t* out) {
  if (b == e) return -1;
  if (*b == '+') {
    b++;
    if (b == e) return -1;
  }
  uint64_t v = 0;
  for (const char* p = b; p < e; ++p) {
    if (*p < '0' || *p > '9') return -1;
    v = v * 10 + (uint64_t)(*p - '0');
    if (v > 0xFFFFFFFFull) return -1;
  }
  *out = (uint32_t)v;
  return 0;
}

static int key_is(const char* b, const char* e, const char* lit) {
  size_t n = (size_t)(e - b);
  return strlen(lit) == n && memcmp(b, lit, n) == 0;
}

static void set_err(char* err, size_t errlen, const char* msg) {
  if (err && errlen) snprintf(err, errlen, "%s", msg);
}

/* ------------------------------------------------------------------ parser
 * shared/src/models/node.rs:180-374 */

static int gpu_req_has_any(const orc_gpu_req* g) { return g->flags != 0; } /* :361-368 */

int orc_parse_requirements(const char* s, orc_requirements* out, char* err, size_t errlen) {
  memset(out, 0, sizeof(*out));
  orc_gpu_req cur;
  memset(&cur, 0, sizeof(cur));
  int started = 0; /* gpu_spec_started :185 */
  const char* end = s + strlen(s);
  const char* p = s;
  while (p <= end) { /* for part in s.split(';') :187 */
    const char* q = memchr(p, ';', (size_t)(end - p));
    if (!q) q = end;
    const char *b = p, *e = q;
    p = q + 1;
    trim(&b, &e); /* :188 */
    if (b == e) continue; /* :189-191 */
    const char* eq = memchr(b, '=', (size_t)(e - b)); /* splitn(2,'=') :193 */
    if (!eq) {
      set_err(err, errlen, "Invalid key-value pair format");
      return 1; /* :194-196 */
    }
    const char *kb = b, *ke = eq, *vb = eq + 1, *ve = e;
    trim(&kb, &ke);
    trim(&vb, &ve);
    uint32_t v = 0;
    int pv = parse_u32(vb, ve, &v);

    if (key_is(kb, ke, "gpu:count")) { /* :203-216 */
      if (started && (cur.flags & ORC_G_COUNT)) {
        if (out->n_gpu >= ORC_MAX_ALTS) {
          set_err(err, errlen, "oracle capacity: too many GPU alternatives");
          return 3;
        }
        out->gpu[out->n_gpu++] = cur;
        memset(&cur, 0, sizeof(cur));
      }
      started = 1;
      if (pv) {
        set_err(err, errlen, "Invalid gpu:count value");
        return 1;
      }
      cur.flags |= ORC_G_COUNT;
      cur.count = v;
    } else if (key_is(kb, ke, "gpu:model")) { /* :217-222 */
      started = 1;
      size_t n = (size_t)(ve - vb);
      if (n >= ORC_MODEL_LEN) {
        set_err(err, errlen, "oracle capacity: model string too long");
        return 3;
      }
      memset(cur.model, 0, sizeof(cur.model));
      memcpy(cur.model, vb, n);
      cur.flags |= ORC_G_MODEL;
    } else if (key_is(kb, ke, "gpu:memory_mb")) { /* :223-238 */
      started = 1;
      if (cur.flags & (ORC_G_MEM_MIN | ORC_G_MEM_MAX)) {
        set_err(err, errlen, "Cannot specify both exact memory and min/max memory");
        return 1;
      }
      if (pv) {
        set_err(err, errlen, "Invalid gpu:memory_mb value");
        return 1;
      }
      cur.flags |= ORC_G_MEM;
      cur.memory_mb = v;
    } else if (key_is(kb, ke, "gpu:memory_mb_min")) { /* :239-262 */
      started = 1;
      if (cur.flags & ORC_G_MEM) {
        set_err(err, errlen, "Cannot specify both exact memory and min/max memory");
        return 1;
      }
      if (cur.flags & ORC_G_MEM_MAX) {
        if (pv) {
          set_err(err, errlen, "panic: unwrap on invalid gpu:memory_mb_min");
          return 2; /* value.parse::<u32>().unwrap() :251 */
        }
        if (cur.memory_mb_max < v) {
          set_err(err, errlen, "min value is greater than max value");
          return 1;
        }
      }
      if (pv) {
        set_err(err, errlen, "Invalid gpu:memory_mb_min value");
        return 1;
      }
      cur.flags |= ORC_G_MEM_MIN;
      cur.memory_mb_min = v;
    } else if (key_is(kb, ke, "gpu:memory_mb_max")) { /* :263-288 */
      started = 1;
      if (cur.flags & ORC_G_MEM) {
        set_err(err, errlen, "Cannot specify both exact memory and min/max memory");
        return 1;
      }
      if (cur.flags & ORC_G_MEM_MIN) {
        if (pv) {
          set_err(err, errlen, "panic: unwrap on invalid gpu:memory_mb_max");
          return 2; /* :275 */
        }
        if (cur.memory_mb_min > v) {
          set_err(err, errlen, "max value is less than min value");
          return 1;
        }
      }
      if (pv) {
        set_err(err, errlen, "Invalid gpu:memory_mb_max value");
        return 1;
      }
      cur.flags |= ORC_G_MEM_MAX;
      cur.memory_mb_max = v;
    } else if (key_is(kb, ke, "gpu:total_memory_min")) { /* :290-310 */
      started = 1;
      if (cur.flags & ORC_G_TOT_MAX) {
        if (pv) {
          set_err(err, errlen, "panic: unwrap on invalid gpu:total_memory_min");
          return 2; /* :296 */
        }
        if (cur.total_memory_max < v) {
          set_err(err, errlen, "min value is greater than max value");
          return 1;
        }
      }
      if (pv) {
        set_err(err, errlen, "Invalid gpu:total_memory_min value");
        return 1;
      }
      cur.flags |= ORC_G_TOT_MIN;
      cur.total_memory_min = v;
    } else if (key_is(kb, ke, "gpu:total_memory_max")) { /* :311-331 */
      started = 1;
      if (cur.flags & ORC_G_TOT_MIN) {
        if (pv) {
          set_err(err, errlen, "panic: unwrap on invalid gpu:total_memory_max");
          return 2; /* :317 */
        }
        if (cur.total_memory_min > v) {
          set_err(err, errlen, "max value is less than min value");
          return 1;
        }
      }
      if (pv) {
        set_err(err, errlen, "Invalid gpu:total_memory_max value");
        return 1;
      }
      cur.flags |= ORC_G_TOT_MAX;
      cur.total_memory_max = v;
    } else if (key_is(kb, ke, "cpu:cores")) { /* :333-341 */
      if (pv) {
        set_err(err, errlen, "Invalid cpu:cores value");
        return 1;
      }
      out->flags |= ORC_R_CPU | ORC_R_CPU_CORES;
      out->cpu_cores = v;
    } else if (key_is(kb, ke, "ram_mb")) { /* :344-350 */
      if (pv) {
        set_err(err, errlen, "Invalid ram_mb value");
        return 1;
      }
      out->flags |= ORC_R_RAM;
      out->ram_mb = v;
    } else if (key_is(kb, ke, "storage_gb")) { /* :351-357 */
      if (pv) {
        set_err(err, errlen, "Invalid storage_gb value");
        return 1;
      }
      out->flags |= ORC_R_STORAGE;
      out->storage_gb = v;
    } else {
      set_err(err, errlen, "Unknown requirement key"); /* :358 */
      return 1;
    }
  }
  if (started && gpu_req_has_any(&cur)) { /* :360-370 */
    if (out->n_gpu >= ORC_MAX_ALTS) {
      set_err(err, errlen, "oracle capacity: too many GPU alternatives");
      return 3;
    }
    out->gpu[out->n_gpu++] = cur;
  }
  return 0;
}

/* ------------------------------------------------------------------ predicate
 * shared/src/models/node.rs:377-541 */

/* to_lowercase().replace(' ', "_") (:465, :470); out holds ORC_MODEL_LEN * 2 bytes */
static void norm_model(const char* b, const char* e, char* out) {
  orc_to_lowercase(b, e, out, ORC_MODEL_LEN * 2);
  for (char* p = out; *p; ++p)
    if (*p == ' ') *p = '_';
}
static void strip_underscore(const char* in, char* out) {
  size_t n = 0;
  for (; *in; ++in)
    if (*in != '_') out[n++] = *in;
  out[n] = 0;
}

void orc_to_lowercase_str(const char* in, char* out, size_t cap) { orc_to_lowercase(in, in + strlen(in), out, cap); }

int orc_model_matches(const char* spec_model, const char* req_model) { /* :463-484 */
  char ns[ORC_MODEL_LEN * 2], ns_nu[ORC_MODEL_LEN * 2];
  norm_model(spec_model, spec_model + strlen(spec_model), ns); /* :465 */
  strip_underscore(ns, ns_nu);                                  /* :473 */
  const char* end = req_model + strlen(req_model);
  const char* p = req_model;
  while (p <= end) { /* req_model.split(',') :468-470 */
    const char* q = memchr(p, ',', (size_t)(end - p));
    if (!q) q = end;
    const char *b = p, *e = q;
    p = q + 1;
    trim(&b, &e);
    char nr[ORC_MODEL_LEN * 2], nr_nu[ORC_MODEL_LEN * 2];
    norm_model(b, e, nr);
    strip_underscore(nr, nr_nu); /* :474 */
    if (strstr(ns, nr) || strstr(nr, ns) || strstr(ns_nu, nr_nu) || strstr(nr_nu, ns_nu)) /* :476-479 */
      return 1;
  }
  return 0;
}

static int gpu_meets(const orc_specs* s, const orc_gpu_req* r) { /* GpuSpecs::meets :445-526 */
  if (r->flags & ORC_G_COUNT) { /* :447-461 */
    if (!(s->flags & ORC_S_G_COUNT)) {
      if (r->count > 0) return 0;
    } else if (s->gpu_count != r->count) {
      return 0;
    }
  }
  if (r->flags & ORC_G_MODEL) { /* :463-484 */
    if (!(s->flags & ORC_S_G_MODEL)) return 0;
    if (!orc_model_matches(s->gpu_model, r->model)) return 0;
  }
  int mem_some = (s->flags & ORC_S_G_MEM) != 0;
  if (r->flags & ORC_G_MEM) /* :487-491 */
    if (!mem_some || s->gpu_memory_mb < r->memory_mb) return 0;
  if (r->flags & ORC_G_MEM_MIN) /* :494-498 */
    if (!mem_some || s->gpu_memory_mb < r->memory_mb_min) return 0;
  if (r->flags & ORC_G_MEM_MAX) /* :499-503 */
    if (!mem_some || s->gpu_memory_mb > r->memory_mb_max) return 0;
  if ((r->flags & ORC_G_TOT_MIN) && (s->flags & ORC_S_G_COUNT) && mem_some) { /* :506-513 */
    uint32_t total = s->gpu_count * s->gpu_memory_mb;                         /* u32 mul, wraps in release */
    if (total < r->total_memory_min) return 0;
  }
  if ((r->flags & ORC_G_TOT_MAX) && (s->flags & ORC_S_G_COUNT) && mem_some) { /* :515-522 */
    uint32_t total = s->gpu_count * s->gpu_memory_mb;
    if (total > r->total_memory_max) return 0;
  }
  return 1;
}

int orc_meets(const orc_specs* s, const orc_requirements* r) { /* ComputeSpecs::meets :379-440 */
  if (r->flags & ORC_R_CPU) { /* :381-393 + CpuSpecs::meets :531-540 */
    if (!(s->flags & ORC_S_CPU)) return 0;
    if (r->flags & ORC_R_CPU_CORES)
      if (!(s->flags & ORC_S_CPU_CORES) || s->cpu_cores < r->cpu_cores) return 0;
  }
  if (r->flags & ORC_R_RAM) /* :396-404 */
    if (!(s->flags & ORC_S_RAM) || s->ram_mb < r->ram_mb) return 0;
  if (r->flags & ORC_R_STORAGE) /* :407-418 */
    if (!(s->flags & ORC_S_STORAGE) || s->storage_gb < r->storage_gb) return 0;
  if (r->n_gpu) { /* :420-435 */
    if (!(s->flags & ORC_S_GPU)) return 0;
    int any = 0;
    for (uint32_t i = 0; i < r->n_gpu && !any; ++i) any = gpu_meets(s, &r->gpu[i]);
    if (!any) return 0;
  }
  return 1;
}

int orc_is_node_compatible_with_config(const orc_config* cfg, const orc_node* node) {
  /* orchestrator/src/plugins/node_groups/mod.rs:206-215 */
  if (!cfg->has_requirements) return 1;        /* (None, _) => true */
  if (!node->has_specs) return 0;              /* (Some, None) => false */
  return orc_meets(&node->specs, &cfg->req);   /* (Some, Some) => meets */
}

/* ------------------------------------------------------------------ Haversine
 * orchestrator/src/plugins/node_groups/mod.rs:218-231 */
double orc_calculate_distance(double lat1, double lon1, double lat2, double lon2) {
  const double EARTH_RADIUS_KM = 6371.0;
  const double RAD = 3.14159265358979323846 / 180.0; /* f64::to_radians: self * (PI/180) */
  double lat1_rad = lat1 * RAD;
  double lat2_rad = lat2 * RAD;
  double delta_lat = (lat2 - lat1) * RAD;
  double delta_lon = (lon2 - lon1) * RAD;
  double s1 = sin(delta_lat / 2.0);
  double s2 = sin(delta_lon / 2.0);
  double a = s1 * s1 + cos(lat1_rad) * cos(lat2_rad) * (s2 * s2); /* powi(2) == x*x */
  double c = 2.0 * atan2(sqrt(a), sqrt(1.0 - a));
  return EARTH_RADIUS_KM * c;
}

/* the same function over a column (test models that need many distances to one reference point) */
void orc_distance_column(double lat0, double lon0, const double* lat, const double* lon, size_t n, double* out) {
  for (size_t i = 0; i < n; ++i) out[i] = orc_calculate_distance(lat0, lon0, lat[i], lon[i]);
}

/* ------------------------------------------------------------------ config ordering */

int orc_sort_configs(const orc_config* cfgs, size_t n, uint32_t* order) { /* mod.rs:138-164 */
  for (size_t i = 0; i < n; ++i) {
    for (size_t j = 0; j < i; ++j)
      if (strncmp(cfgs[i].name, cfgs[j].name, ORC_NAME_LEN) == 0) return 2; /* :142-144 */
    if (cfgs[i].max_group_size < cfgs[i].min_group_size) return 2;            /* :145-147 */
  }
  for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
  /* stable insertion sort with the comparator of :150-164 */
  for (size_t i = 1; i < n; ++i) {
    uint32_t x = order[i];
    size_t j = i;
    while (j > 0) {
      const orc_config* a = &cfgs[order[j - 1]];
      const orc_config* b = &cfgs[x];
      int cmp; /* cmp(a,b): <0 a first */
      if (a->min_group_size != b->min_group_size)
        cmp = (b->min_group_size < a->min_group_size) ? -1 : 1; /* desc */
      else if (a->has_requirements && !b->has_requirements)
        cmp = -1;
      else if (!a->has_requirements && b->has_requirements)
        cmp = 1;
      else
        cmp = 0;
      if (cmp <= 0) break; /* stable: move only when strictly greater */
      order[j] = order[j - 1];
      --j;
    }
    order[j] = x;
  }
  return 0;
}

size_t orc_available_configs(const orc_config* cfgs, const uint32_t* template_order, size_t n,
                             const uint8_t* enabled, uint32_t* out) { /* mod.rs:399-418 */
  size_t m = 0;
  for (size_t i = 0; i < n; ++i)
    if (enabled[template_order[i]]) out[m++] = template_order[i]; /* :409-414 */
  for (size_t i = 1; i < m; ++i) {                                /* :416 stable, min desc only */
    uint32_t x = out[i];
    size_t j = i;
    while (j > 0 && cfgs[out[j - 1]].min_group_size < cfgs[x].min_group_size) {
      out[j] = out[j - 1];
      --j;
    }
    out[j] = x;
  }
  return m;
}

/* ------------------------------------------------------------------ task side */

int64_t orc_newest_task(const orc_task* tasks, size_t n) { /* newest_task/mod.rs:8-19 */
  if (n == 0) return -1;
  size_t best = 0;
  for (size_t i = 1; i < n; ++i)
    if (tasks[i].created_at >= tasks[best].created_at) best = i; /* max_by_key keeps the LAST max */
  return (int64_t)best;
}

int orc_task_applicable(const orc_task* t, const char* config_name) {
  /* scheduler_impl.rs:44-59, mod.rs:1138-1162 */
  if (!t->restricted) return 1;
  for (uint32_t i = 0; i < t->n_topologies; ++i)
    if (strncmp(t->topologies[i], config_name, ORC_NAME_LEN) == 0) return 1; /* Vec<String>::contains */
  return 0;
}

/* ------------------------------------------------------------------ state */

typedef struct {
  int alive;
  uint64_t id;
  char id_str[20]; /* format!("{:x}", u64) */
  uint32_t cfg;    /* index into cfgs */
  uint32_t n;
  uint32_t* members; /* node indices in address order (BTreeSet<String>) */
  int64_t task;      /* group_task:<id> or -1 */
} grp;

struct orc_state {
  const orc_node* nodes;
  size_t n_nodes;
  const orc_config* cfgs;
  size_t n_cfgs;
  orc_policy pol;
  uint32_t* template_order;
  uint8_t* enabled;
  const orc_task* tasks;
  size_t n_tasks;
  int32_t* node_group;
  grp* groups;
  size_t n_slots, cap_slots, n_live;
  uint64_t id_rng;
  uint64_t n_hav, n_meets;
  /* webhook feed: what send_group_created / send_group_destroyed would have been called with, in call order
   * (mod.rs:612-625, 974-1000, 1469-1481) */
  orc_group_event* ev;
  size_t n_ev, cap_ev;
  uint32_t* ev_mem;
  size_t n_evm, cap_evm;
};

static void log_event(orc_state* st, uint32_t kind, const grp* g) {
  if (st->n_ev == st->cap_ev) {
    st->cap_ev = st->cap_ev ? st->cap_ev * 2 : 256;
    st->ev = realloc(st->ev, st->cap_ev * sizeof(*st->ev));
  }
  while (st->n_evm + g->n > st->cap_evm) {
    st->cap_evm = st->cap_evm ? st->cap_evm * 2 : 1024;
    st->ev_mem = realloc(st->ev_mem, st->cap_evm * sizeof(uint32_t));
  }
  orc_group_event* e = &st->ev[st->n_ev++];
  e->group_id = g->id;
  e->kind = kind;
  e->config = g->cfg;
  e->member_begin = (uint32_t)st->n_evm;
  e->n_members = g->n;
  memcpy(st->ev_mem + st->n_evm, g->members, sizeof(uint32_t) * g->n); /* group.nodes.iter(): BTreeSet order */
  st->n_evm += g->n;
}

size_t orc_events(const orc_state* st, orc_group_event* out, size_t cap, uint32_t* members, size_t cap_members,
                  size_t* n_members) {
  if (n_members) *n_members = st->n_evm;
  if (out && cap >= st->n_ev) memcpy(out, st->ev, st->n_ev * sizeof(*out));
  if (members && cap_members >= st->n_evm) memcpy(members, st->ev_mem, st->n_evm * sizeof(uint32_t));
  return st->n_ev;
}
void orc_events_clear(orc_state* st) { st->n_ev = st->n_evm = 0; }

orc_state* orc_state_new(const orc_node* nodes, size_t n_nodes, const orc_config* cfgs, size_t n_cfgs,
                         const orc_policy* policy) {
  orc_state* st = calloc(1, sizeof(*st));
  st->nodes = nodes;
  st->n_nodes = n_nodes;
  st->cfgs = cfgs;
  st->n_cfgs = n_cfgs;
  st->pol = *policy;
  st->template_order = malloc(sizeof(uint32_t) * (n_cfgs ? n_cfgs : 1));
  if (orc_sort_configs(cfgs, n_cfgs, st->template_order) != 0) {
    free(st->template_order);
    free(st);
    return NULL;
  }
  st->enabled = calloc(n_cfgs ? n_cfgs : 1, 1);
  st->node_group = malloc(sizeof(int32_t) * (n_nodes ? n_nodes : 1));
  for (size_t i = 0; i < n_nodes; ++i) st->node_group[i] = -1;
  st->id_rng = policy->group_id_seed;
  return st;
}

void orc_state_free(orc_state* st) {
  if (!st) return;
  for (size_t i = 0; i < st->n_slots; ++i) free(st->groups[i].members);
  free(st->ev);
  free(st->ev_mem);
  free(st->groups);
  free(st->node_group);
  free(st->enabled);
  free(st->template_order);
  free(st);
}

void orc_state_set_enabled(orc_state* st, const uint8_t* enabled) { memcpy(st->enabled, enabled, st->n_cfgs); }
void orc_state_set_tasks(orc_state* st, const orc_task* tasks, size_t n) {
  st->tasks = tasks;
  st->n_tasks = n;
}

static int addr_cmp(const orc_state* st, uint32_t a, uint32_t b) {
  return strcmp(st->nodes[a].address, st->nodes[b].address);
}

/* create_group_atomically (mod.rs:299-322) / execute_group_merge's create half (:927-942) */
static uint32_t new_group(orc_state* st, uint32_t cfg, const uint32_t* members, uint32_t n) {
  if (st->n_slots == st->cap_slots) {
    st->cap_slots = st->cap_slots ? st->cap_slots * 2 : 64;
    st->groups = realloc(st->groups, st->cap_slots * sizeof(grp));
  }
  grp* g = &st->groups[st->n_slots];
  g->alive = 1;
  g->id = orc_splitmix64(&st->id_rng); /* generate_group_id :1489-1493, injected */
  snprintf(g->id_str, sizeof(g->id_str), "%llx", (unsigned long long)g->id);
  g->cfg = cfg;
  g->n = n;
  g->task = -1;
  g->members = malloc(sizeof(uint32_t) * (n ? n : 1));
  memcpy(g->members, members, sizeof(uint32_t) * n);
  for (uint32_t i = 1; i < n; ++i) { /* BTreeSet<String>: address byte order */
    uint32_t x = g->members[i];
    uint32_t j = i;
    while (j > 0 && addr_cmp(st, g->members[j - 1], x) > 0) {
      g->members[j] = g->members[j - 1];
      --j;
    }
    g->members[j] = x;
  }
  for (uint32_t i = 0; i < n; ++i) st->node_group[members[i]] = (int32_t)st->n_slots;
  st->n_live++;
  log_event(st, ORC_GROUP_CREATED, g);
  return (uint32_t)st->n_slots++;
}

void orc_dissolve_group(orc_state* st, uint32_t slot) { /* mod.rs:1423-1487 */
  if (slot >= st->n_slots || !st->groups[slot].alive) return;
  grp* g = &st->groups[slot];
  log_event(st, ORC_GROUP_DESTROYED, g); /* :1469-1481 */
  for (uint32_t i = 0; i < g->n; ++i)
    if (st->node_group[g->members[i]] == (int32_t)slot) st->node_group[g->members[i]] = -1;
  g->alive = 0;
  g->task = -1;
  st->n_live--;
}

void orc_state_set_node_status(orc_state* st, orc_node* nodes_mut, size_t idx, uint32_t status) {
  nodes_mut[idx].status = status;
  /* status_update_impl.rs:17-29 */
  if ((status == ORC_ST_DEAD || status == ORC_ST_LOWBALANCE) && st->node_group[idx] >= 0)
    orc_dissolve_group(st, (uint32_t)st->node_group[idx]);
}

void orc_state_remap_tasks(orc_state* st, const int64_t* map, size_t n_old) {
  for (size_t g = 0; g < st->n_slots; ++g) {
    grp* gr = &st->groups[g];
    if (!gr->alive || gr->task < 0) continue;
    int64_t nt = (size_t)gr->task < n_old ? map[gr->task] : -1;
    if (nt < 0)
      orc_dissolve_group(st, (uint32_t)g); /* mod.rs:1259-1288 */
    else
      gr->task = nt;
  }
}

/* ---- stable merge sort over uint32 index arrays with a context comparator (Rust sort_by is stable;
 * any stable sort produces the same permutation for a consistent comparator). */
typedef int (*cmp_fn)(void* ctx, uint32_t a, uint32_t b);
static void msort(uint32_t* a, uint32_t* tmp, size_t n, cmp_fn cmp, void* ctx) {
  if (n < 2) return;
  size_t h = n / 2;
  msort(a, tmp, h, cmp, ctx);
  msort(a + h, tmp, n - h, cmp, ctx);
  size_t i = 0, j = h, k = 0;
  while (i < h && j < n) {
    if (cmp(ctx, a[j], a[i]) < 0)
      tmp[k++] = a[j++];
    else
      tmp[k++] = a[i++];
  }
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, n * sizeof(uint32_t));
}

typedef struct {
  orc_state* st;
  const orc_node* ref;
} prox_ctx;

/* sort_nodes_by_proximity comparator, mod.rs:239-253: two Haversines per comparison. */
static int prox_cmp(void* vctx, uint32_t a, uint32_t b) {
  prox_ctx* c = vctx;
  const orc_node* na = &c->st->nodes[a];
  const orc_node* nb = &c->st->nodes[b];
  double da = 1.7976931348623157e308, db = 1.7976931348623157e308; /* f64::MAX */
  if (na->has_location) {
    da = orc_calculate_distance(c->ref->latitude, c->ref->longitude, na->latitude, na->longitude);
    c->st->n_hav++;
  }
  if (nb->has_location) {
    db = orc_calculate_distance(c->ref->latitude, c->ref->longitude, nb->latitude, nb->longitude);
    c->st->n_hav++;
  }
  if (da < db) return -1;
  if (da > db) return 1;
  return 0; /* Equal, and partial_cmp None (NaN) => Equal */
}

/* cached-distance comparator for the best-effort variant (ctx = keys indexed by node) */
static int key_cmp(void* ctx, uint32_t a, uint32_t b) {
  const double* k = ctx;
  if (k[a] < k[b]) return -1;
  if (k[a] > k[b]) return 1;
  return 0;
}

size_t orc_try_form_new_groups(orc_state* st) { /* mod.rs:478-628 */
  size_t formed = 0;
  st->n_hav = st->n_meets = 0;
  size_t W = st->n_nodes;
  uint32_t* avail = malloc(sizeof(uint32_t) * (st->n_cfgs ? st->n_cfgs : 1));
  size_t n_avail = orc_available_configs(st->cfgs, st->template_order, st->n_cfgs, st->enabled, avail); /* :483 */

  uint32_t* healthy = malloc(sizeof(uint32_t) * (W ? W : 1)); /* :492-497 */
  size_t n_healthy = 0;
  for (size_t i = 0; i < W; ++i)
    if (st->nodes[i].status == ORC_ST_HEALTHY && st->nodes[i].has_p2p && st->node_group[i] < 0)
      healthy[n_healthy++] = (uint32_t)i;
  uint32_t* compat = malloc(sizeof(uint32_t) * (W ? W : 1));
  uint32_t* tmp = malloc(sizeof(uint32_t) * (W ? W : 1));
  uint32_t* members = malloc(sizeof(uint32_t) * (W ? W : 1));
  uint8_t* in_group = calloc(W ? W : 1, 1);
  double* keys = malloc(sizeof(double) * (W ? W : 1));
  uint8_t* mask_once = NULL;

  size_t total_available = n_healthy; /* :503 */
  for (size_t ci = 0; ci < n_avail; ++ci) { /* :505 */
    const orc_config* cfg = &st->cfgs[avail[ci]];
    if (!st->pol.reference_shaped) { /* best-effort: predicate once per (config, node) */
      if (!mask_once) mask_once = malloc(W ? W : 1);
      for (size_t k = 0; k < n_healthy; ++k) {
        mask_once[healthy[k]] = (uint8_t)orc_is_node_compatible_with_config(cfg, &st->nodes[healthy[k]]);
        st->n_meets++;
      }
    }
    while (total_available >= cfg->min_group_size) { /* :507 */
      size_t initial_available = total_available;
      size_t n_compat = 0; /* :511-515 */
      for (size_t k = 0; k < n_healthy; ++k) {
        int ok;
        if (st->pol.reference_shaped) {
          ok = orc_is_node_compatible_with_config(cfg, &st->nodes[healthy[k]]);
          st->n_meets++;
        } else {
          ok = mask_once[healthy[k]];
        }
        if (ok) compat[n_compat++] = healthy[k];
      }
      if (n_compat < cfg->min_group_size) break; /* :517-519 */

      size_t n_members = 0;
      if (st->pol.proximity_enabled) { /* :524 */
        int have_seed = 0;
        uint32_t seed = 0;
        for (size_t k = 0; k < n_compat && !have_seed; ++k) /* :526-530 find(location.is_some()) */
          if (st->nodes[compat[k]].has_location) {
            seed = compat[k];
            have_seed = 1;
          }
        if (!have_seed && n_compat) { /* .or(first()) */
          seed = compat[0];
          have_seed = 1;
        }
        if (have_seed) {
          members[n_members++] = seed; /* :534-535 */
          size_t n_rem = 0;            /* :538-541 filter address != seed.address */
          for (size_t k = 0; k < n_compat; ++k)
            if (compat[k] != seed) compat[n_rem++] = compat[k];
          const orc_node* sn = &st->nodes[seed];
          if (sn->has_location) { /* sort_nodes_by_proximity :238 */
            if (st->pol.reference_shaped) {
              prox_ctx ctx = {st, sn};
              msort(compat, tmp, n_rem, prox_cmp, &ctx);
            } else {
              for (size_t k = 0; k < n_rem; ++k) {
                const orc_node* nk = &st->nodes[compat[k]];
                if (nk->has_location) {
                  keys[compat[k]] = orc_calculate_distance(sn->latitude, sn->longitude, nk->latitude, nk->longitude);
                  st->n_hav++;
                } else {
                  keys[compat[k]] = 1.7976931348623157e308;
                }
              }
              msort(compat, tmp, n_rem, key_cmp, keys);
            }
          }
          for (size_t k = 0; k < n_rem; ++k) { /* :545-551 */
            if (n_members >= cfg->max_group_size) break;
            members[n_members++] = compat[k];
          }
        }
      } else { /* :553-561 */
        for (size_t k = 0; k < n_compat; ++k) {
          if (n_members >= cfg->max_group_size) break;
          members[n_members++] = compat[k];
        }
      }
      if (n_members < cfg->min_group_size) break; /* :564-566 */

      new_group(st, avail[ci], members, (uint32_t)n_members); /* :569-581 */
      formed++;
      for (size_t k = 0; k < n_members; ++k) in_group[members[k]] = 1; /* :585 retain */
      size_t m = 0;
      for (size_t k = 0; k < n_healthy; ++k)
        if (!in_group[healthy[k]]) healthy[m++] = healthy[k];
      n_healthy = m;
      total_available = n_healthy;                   /* :586 */
      if (total_available == initial_available) break; /* :608-610 */
    }
  }
  free(mask_once);
  free(keys);
  free(in_group);
  free(members);
  free(tmp);
  free(compat);
  free(healthy);
  free(avail);
  return formed;
}

/* ---- chooser: injected replacement for rand::rng() + IteratorRandom::choose
 * (scheduler_impl.rs:66-70, mod.rs:1175-1177). Returns the rank (0-based) within the applicable list. */
static uint64_t choose_rank(const orc_state* st, uint64_t group_id, uint64_t n_applicable) {
  if (st->pol.chooser == ORC_CHOOSE_SEEDED) return mix64(st->pol.chooser_seed ^ group_id) % n_applicable;
  return 0;
}

/* applicable-task filter + choose for a configuration name. -1 if none. */
static int64_t pick_task_for(const orc_state* st, const char* config_name, uint64_t group_id) {
  uint64_t n_app = 0;
  for (size_t t = 0; t < st->n_tasks; ++t) n_app += (uint64_t)orc_task_applicable(&st->tasks[t], config_name);
  if (n_app == 0) return -1; /* scheduler_impl.rs:62-64, mod.rs:1166-1172 */
  uint64_t r = choose_rank(st, group_id, n_app);
  for (size_t t = 0; t < st->n_tasks; ++t)
    if (orc_task_applicable(&st->tasks[t], config_name)) {
      if (r == 0) return (int64_t)t;
      --r;
    }
  return -1;
}

/* get_all_groups (mod.rs:1006-1044): live groups sorted by id string. */
static size_t all_groups_sorted(const orc_state* st, uint32_t* out) {
  size_t n = 0;
  for (size_t i = 0; i < st->n_slots; ++i)
    if (st->groups[i].alive) out[n++] = (uint32_t)i;
  for (size_t i = 1; i < n; ++i) { /* :1040 sort_by id.cmp */
    uint32_t x = out[i];
    size_t j = i;
    while (j > 0 && strcmp(st->groups[out[j - 1]].id_str, st->groups[x].id_str) > 0) {
      out[j] = out[j - 1];
      --j;
    }
    out[j] = x;
  }
  return n;
}

static int merge_dist_cmp(void* ctx, uint32_t a, uint32_t b) {
  const double* d = ctx;
  if (d[a] < d[b]) return -1;
  if (d[a] > d[b]) return 1;
  return 0;
}

/* attempt_group_merge (mod.rs:752-860) + is_merge_beneficial (:863-873) + should_switch_tasks
 * (:257-296) + execute_group_merge (:876-971).  `rem` = remaining compatible solo group slots in
 * get_all_groups order.  On success writes the used slots to used[0..*n_used) and returns 1. */
static int attempt_group_merge(orc_state* st, const uint32_t* rem, size_t n_rem, uint32_t cfg_idx,
                               uint32_t* used, size_t* n_used) {
  const orc_config* cfg = &st->cfgs[cfg_idx];
  size_t nb = 0; /* merge_batch == used; total_nodes == nb (all solo) */
  if (st->pol.proximity_enabled) { /* :762 */
    int have_seed = 0;
    size_t seed_pos = 0;
    for (size_t i = 0; i < n_rem && !have_seed; ++i) /* :772-781 */
      if (st->nodes[st->groups[rem[i]].members[0]].has_location) {
        seed_pos = i;
        have_seed = 1;
      }
    if (have_seed) {
      const orc_node* sn = &st->nodes[st->groups[rem[seed_pos]].members[0]];
      used[nb++] = rem[seed_pos]; /* :787-789 */
      uint32_t* idx = malloc(sizeof(uint32_t) * (n_rem ? n_rem : 1));
      uint32_t* tmp = malloc(sizeof(uint32_t) * (n_rem ? n_rem : 1));
      double* d = malloc(sizeof(double) * (n_rem ? n_rem : 1));
      size_t m = 0;
      for (size_t i = 0; i < n_rem; ++i) { /* :792-804 filter_map: only groups whose node has a location */
        if (i == seed_pos) continue;
        const orc_node* nk = &st->nodes[st->groups[rem[i]].members[0]];
        if (!nk->has_location) continue;
        d[i] = orc_calculate_distance(sn->latitude, sn->longitude, nk->latitude, nk->longitude);
        st->n_hav++;
        idx[m++] = (uint32_t)i;
      }
      msort(idx, tmp, m, merge_dist_cmp, d); /* :806-807 stable */
      for (size_t k = 0; k < m; ++k) {       /* :810-820 */
        if (nb + 1 <= cfg->max_group_size) {
          used[nb++] = rem[idx[k]];
          if (nb >= cfg->max_group_size) break;
        }
      }
      free(d);
      free(tmp);
      free(idx);
    }
  }
  if (nb == 0 || (nb < cfg->max_group_size && nb < cfg->min_group_size)) { /* :824-827 */
    if (nb < cfg->min_group_size) nb = 0;                                   /* :829-833 */
    for (size_t i = 0; i < n_rem; ++i) {                                    /* :836-848 */
      int already = 0;
      for (size_t k = 0; k < nb; ++k) already |= (used[k] == rem[i]);
      if (!already && nb + 1 <= cfg->max_group_size) {
        used[nb++] = rem[i];
        if (nb >= cfg->max_group_size) break;
      }
    }
  }
  /* is_merge_beneficial :863-873 */
  if (nb < 2) return 0;
  /* should_switch_tasks :257-296 (all groups here are solo, potential size == nb >= 2) */
  if (!st->pol.switching_enabled) return 0;
  if (!st->pol.prefer_larger_groups)
    for (size_t k = 0; k < nb; ++k)
      if (st->groups[used[k]].task >= 0) return 0; /* :277-287 */

  /* execute_group_merge :876-971 */
  uint32_t* mem = malloc(sizeof(uint32_t) * nb);
  for (size_t k = 0; k < nb; ++k) mem[k] = st->groups[used[k]].members[0];
  for (size_t k = 0; k < nb; ++k) orc_dissolve_group(st, used[k]); /* :903-921 */
  uint32_t slot = new_group(st, cfg_idx, mem, (uint32_t)nb);       /* :886-892, :924-936 */
  grp* g = &st->groups[slot];
  g->task = pick_task_for(st, cfg->name, g->id); /* find_best_task_for_group :896, SETNX :939-942 */
  free(mem);
  *n_used = nb;
  return 1;
}

size_t orc_try_merge_solo_groups(orc_state* st) { /* mod.rs:631-673 */
  size_t merged = 0;
  uint32_t* all = malloc(sizeof(uint32_t) * (st->n_slots ? st->n_slots : 1));
  size_t n_all = all_groups_sorted(st, all);
  size_t solo = 0;
  for (size_t i = 0; i < n_all; ++i) solo += (st->groups[all[i]].n == 1);
  if (solo < 2) { /* :641-644 */
    free(all);
    return 0;
  }
  uint32_t* avail = malloc(sizeof(uint32_t) * (st->n_cfgs ? st->n_cfgs : 1));
  size_t n_avail = orc_available_configs(st->cfgs, st->template_order, st->n_cfgs, st->enabled, avail);
  for (size_t ci = 0; ci < n_avail; ++ci) { /* :654 */
    uint32_t cfg_idx = avail[ci];
    const orc_config* cfg = &st->cfgs[cfg_idx];
    /* current_groups = get_all_groups() :656; slots may have grown */
    all = realloc(all, sizeof(uint32_t) * (st->n_slots ? st->n_slots : 1));
    n_all = all_groups_sorted(st, all);
    /* find_compatible_solo_groups :712-734 (+ is_group_compatible_with_config :737-749) */
    uint32_t* rem = malloc(sizeof(uint32_t) * (n_all ? n_all : 1));
    size_t n_rem = 0;
    for (size_t i = 0; i < n_all; ++i) {
      const grp* g = &st->groups[all[i]];
      if (g->n == 1 && orc_is_node_compatible_with_config(cfg, &st->nodes[g->members[0]])) rem[n_rem++] = all[i];
    }
    if (n_rem >= cfg->min_group_size) {                  /* :688-691 */
      uint32_t* used = malloc(sizeof(uint32_t) * (n_rem ? n_rem : 1));
      while (n_rem >= cfg->min_group_size) {             /* :695 */
        size_t n_used = 0;
        if (!attempt_group_merge(st, rem, n_rem, cfg_idx, used, &n_used)) break; /* :705 */
        merged++;
        size_t m = 0; /* :702 retain */
        for (size_t i = 0; i < n_rem; ++i) {
          int u = 0;
          for (size_t k = 0; k < n_used; ++k) u |= (used[k] == rem[i]);
          if (!u) rem[m++] = rem[i];
        }
        n_rem = m;
      }
      free(used);
    }
    free(rem);
  }
  free(avail);
  free(all);
  return merged;
}

/* ------------------------------------------------------------------ per-node scheduling */

int64_t orc_filter_tasks_node_groups(orc_state* st, size_t node_idx, uint32_t* group_index,
                                     uint32_t* group_size, uint32_t* next_node) {
  /* scheduler_impl.rs:11-110 */
  int32_t slot = st->node_group[node_idx];
  if (slot < 0) return -1; /* :208-209 not in a group => vec![] */
  grp* g = &st->groups[slot];
  uint32_t idx = 0; /* get_idx_in_group mod.rs:424-434 */
  for (uint32_t i = 0; i < g->n; ++i)
    if (g->members[i] == node_idx) idx = i;
  int64_t cur = g->task; /* get_current_group_task :33 */
  if (cur < 0) {
    if (st->n_tasks == 0) return -1;                                /* :38-40 */
    cur = pick_task_for(st, st->cfgs[g->cfg].name, g->id);         /* :42-70 */
    if (cur < 0) return -1;                                         /* :62-64 */
    g->task = cur; /* assign_task_to_group SETNX :74 — single-threaded, so always Ok(true) */
  }
  if (group_index) *group_index = idx;
  if (group_size) *group_size = g->n;
  if (next_node) *next_node = g->members[(idx + 1) % g->n]; /* :115-116 */
  return cur;
}

int64_t orc_get_task_for_node(orc_state* st, size_t node_idx, int use_node_groups) {
  /* scheduler/mod.rs:26-36: fold the plugin chain, take [0] */
  if (use_node_groups) return orc_filter_tasks_node_groups(st, node_idx, NULL, NULL, NULL);
  return orc_newest_task(st->tasks, st->n_tasks); /* default plugin :16-19 */
}

/* ------------------------------------------------------------------ read-back */

size_t orc_n_groups(const orc_state* st) { return st->n_live; }
size_t orc_group_slots(const orc_state* st) { return st->n_slots; }
const int32_t* orc_node_to_group(const orc_state* st) { return st->node_group; }

int orc_group_info(const orc_state* st, uint32_t slot, uint64_t* id, uint32_t* config_idx,
                   uint32_t* n_members, uint32_t* members, size_t cap, int64_t* task_idx) {
  if (slot >= st->n_slots || !st->groups[slot].alive) return 0;
  const grp* g = &st->groups[slot];
  if (id) *id = g->id;
  if (config_idx) *config_idx = g->cfg;
  if (n_members) *n_members = g->n;
  if (members)
    for (uint32_t i = 0; i < g->n && i < cap; ++i) members[i] = g->members[i];
  if (task_idx) *task_idx = g->task;
  return 1;
}

void orc_counters(const orc_state* st, uint64_t* n_haversine, uint64_t* n_meets) {
  if (n_haversine) *n_haversine = st->n_hav;
  if (n_meets) *n_meets = st->n_meets;
}

/* ------------------------------------------------------------------ whole-table helpers */

void orc_compat_masks(const orc_node* nodes, size_t n_nodes, const orc_config* cfgs, size_t n_cfgs,
                      uint64_t* mask_out) {
  for (size_t w = 0; w < n_nodes; ++w) {
    uint64_t m = 0;
    for (size_t c = 0; c < n_cfgs && c < 64; ++c)
      if (orc_is_node_compatible_with_config(&cfgs[c], &nodes[w])) m |= 1ull << c;
    mask_out[w] = m;
  }
}

void orc_pair_sweep_per_worker(const orc_task* tasks, size_t n_tasks, const orc_config* cfgs,
                               const int32_t* cfg_of_node, size_t n_nodes, uint32_t* first_out,
                               uint32_t* count_out) {
  for (size_t w = 0; w < n_nodes; ++w) {
    uint32_t first = 0xFFFFFFFFu, count = 0;
    if (cfg_of_node[w] >= 0) {
      const char* name = cfgs[cfg_of_node[w]].name;
      for (size_t t = 0; t < n_tasks; ++t) /* scheduler_impl.rs:42-61, one heartbeat */
        if (orc_task_applicable(&tasks[t], name)) {
          if (count == 0) first = (uint32_t)t;
          count++;
        }
    }
    first_out[w] = first;
    count_out[w] = count;
  }
}

/* The same sweep with the nodes split over threads: in the reference the heartbeats of different nodes are
 * independent requests on the tokio worker threads, so this is what all host cores can do for that phase
 * (bench.py: cpu_baseline.all_cores).  Group formation stays sequential, as it is in the reference. */
typedef struct {
  const orc_task* tasks; size_t n_tasks; const orc_config* cfgs; const int32_t* cfg_of_node;
  size_t start, stride, n_nodes; uint32_t* first_out; uint32_t* count_out;
} orc_sweep_job;

static void* orc_sweep_thread(void* arg) { /* nodes start, start + stride, ...: grouped nodes come in runs */
  const orc_sweep_job* j = (const orc_sweep_job*)arg;
  for (size_t w = j->start; w < j->n_nodes; w += j->stride)
    orc_pair_sweep_per_worker(j->tasks, j->n_tasks, j->cfgs, j->cfg_of_node + w, 1, j->first_out + w,
                              j->count_out + w);
  return NULL;
}

int orc_pair_sweep_per_worker_mt(const orc_task* tasks, size_t n_tasks, const orc_config* cfgs,
                                 const int32_t* cfg_of_node, size_t n_nodes, uint32_t* first_out,
                                 uint32_t* count_out, uint32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 1024) n_threads = 1024;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
  orc_sweep_job* jobs = (orc_sweep_job*)malloc(sizeof(orc_sweep_job) * n_threads);
  int rc = 0;
  uint32_t started = 0;
  for (uint32_t i = 0; i < n_threads; ++i) {
    jobs[i] = (orc_sweep_job){tasks, n_tasks, cfgs, cfg_of_node, i, n_threads, n_nodes, first_out, count_out};
    if (pthread_create(&th[i], NULL, orc_sweep_thread, &jobs[i]) != 0) {
      orc_sweep_thread(&jobs[i]); /* run it here */
      th[i] = 0;
      rc = 1;
      continue;
    }
    started |= 1u; /* at least one */
  }
  (void)started;
  for (uint32_t i = 0; i < n_threads; ++i)
    if (th[i]) pthread_join(th[i], NULL);
  free(th);
  free(jobs);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Group variables (scheduler_impl.rs:155-200, storage.rs:150-215): chains of Rust `str::replace`, i.e.
 * every non-overlapping match from the left is replaced and the NEXT replace runs over that result. */

/* one `s.replace(pat, with)`: consumes s (free), returns a new malloc'd string */
static char* orc_replace(char* s, const char* pat, const char* with) {
  const size_t ls = strlen(s), lp = strlen(pat), lw = strlen(with);
  size_t hits = 0;
  for (const char* q = s; (q = strstr(q, pat)) != NULL; q += lp) ++hits;
  char* out = (char*)malloc(ls + hits * (lw > lp ? lw - lp : 0) + 1);
  char* o = out;
  const char* q = s;
  for (;;) {
    const char* h = strstr(q, pat);
    if (!h) break;
    memcpy(o, q, (size_t)(h - q));
    o += h - q;
    memcpy(o, with, lw);
    o += lw;
    q = h + lp;
  }
  strcpy(o, q);
  free(s);
  return out;
}

static char* orc_dup(const char* s) {
  char* r = (char*)malloc(strlen(s) + 1);
  strcpy(r, s);
  return r;
}

/* `total_upload_count.parse::<u32>().unwrap_or(0).saturating_sub(1)` (scheduler_impl.rs:155-158) */
uint32_t orc_last_file_idx(const char* t) {
  size_t i = 0;
  unsigned long long v = 0;
  if (t[0] == '+') i = 1;          /* u32::from_str accepts one leading '+' */
  if (t[i] == '\0') return 0;      /* empty / sign only: Err */
  for (; t[i]; ++i) {
    if (t[i] < '0' || t[i] > '9') return 0; /* Err: InvalidDigit */
    v = v * 10ull + (unsigned long long)(t[i] - '0');
    if (v > 4294967295ull) return 0;        /* Err: PosOverflow */
  }
  return v == 0 ? 0u : (uint32_t)(v - 1ull);
}

char* orc_group_vars(const char* in, uint32_t group_index, uint32_t group_size, const char* next_p2p_address,
                     const char* group_id, const char* total_upload_count) {
  char num[32];
  char* s = orc_dup(in);
  snprintf(num, sizeof num, "%u", group_index);
  s = orc_replace(s, "${GROUP_INDEX}", num);           /* :162, :173 */
  snprintf(num, sizeof num, "%u", group_size);
  s = orc_replace(s, "${GROUP_SIZE}", num);            /* :163, :174 */
  s = orc_replace(s, "${NEXT_P2P_ADDRESS}", next_p2p_address);
  s = orc_replace(s, "${GROUP_ID}", group_id);
  s = orc_replace(s, "${TOTAL_UPLOAD_COUNT}", total_upload_count);
  snprintf(num, sizeof num, "%u", orc_last_file_idx(total_upload_count));
  s = orc_replace(s, "${LAST_FILE_IDX}", num);
  return s;
}

char* orc_volume_vars(const char* in, const char* group_id) { /* :185-200 */
  return orc_replace(orc_dup(in), "${GROUP_ID}", group_id);
}

char* orc_upload_name_vars(const char* in, const char* group_id, uint32_t group_size, uint32_t group_index,
                           uint64_t upload_count) {
  char num[32];
  char* s = orc_dup(in);
  if (group_id) { /* storage.rs:150-158 */
    s = orc_replace(s, "${NODE_GROUP_ID}", group_id);
    snprintf(num, sizeof num, "%u", group_size);
    s = orc_replace(s, "${NODE_GROUP_SIZE}", num);
    snprintf(num, sizeof num, "%u", group_index);
    s = orc_replace(s, "${NODE_GROUP_INDEX}", num);
  }
  snprintf(num, sizeof num, "%llu", (unsigned long long)upload_count);
  s = orc_replace(s, "${TOTAL_UPLOAD_COUNT_AFTER}", num); /* :210-212 (the `contains` guard changes nothing) */
  snprintf(num, sizeof num, "%llu", (unsigned long long)(upload_count ? upload_count - 1 : 0)); /* :208 */
  s = orc_replace(s, "${CURRENT_FILE_INDEX}", num);
  return s;
}

void orc_free_string(char* s) { free(s); }
