// wide_sim.cpp -- round 4 experiment (VERDICT r03 task 2): what would an 8-WIDE compressed node cost the walk?
// (Ylitie, Karras, Laine, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017: 80-byte node =
// origin 12 B + 3 exponent bytes + imask + child base + triangle base + 8 meta bytes + 6 x 8 plane bytes: FIVE 16-byte pieces.)
// Not product code, not linked into libpbrt_hip.so: a simulator that collapses the product's optimised binary tree (canonical
// binned SAH -> single-triangle leaves -> the device builder's re-insertion pass, host run) into W-wide nodes with child boxes on
// the node's own 8-bit power-of-two grid (the product's quantisation rule, capi.cpp make_quad_nodes_as), walks the ray mix of a
// path-traced frame (written by run.py) and counts node steps and triangle tests per ray for W = 4 and W = 8, both with the greedy
// largest-area collapse.  Hits are checked against the oracle's (passed in by run.py).
// build: g++ -O2 -std=c++17 -Ipbrt_amd/csrc tools/experiments/r04_wide8/wide_sim.cpp pbrt_amd/csrc/bvh_build.cpp pbrt_amd/csrc/reinsert_batch.cpp -o /tmp/wide_sim
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bvh_build.hpp"
#include "reinsert_batch.hpp"

using namespace pbrt_hip;

static std::vector<char> slurp(const char *path) {
  FILE *f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(1); }
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)n);
  if (std::fread(b.data(), 1, (size_t)n, f) != (size_t)n) { std::perror("read"); std::exit(1); }
  std::fclose(f);
  return b;
}

constexpr uint32_t kLeaf = 0x80000000u, kDone = 0xffffffffu;
constexpr float kTMin = 1e-4f, kPad = 0x1.000006p+0f;

struct WNode {
  int n = 0;
  float lo[8][3], hi[8][3];  // decoded (quantised, enclosing) child boxes
  uint32_t ref[8];           // kLeaf | reference, or node index
};

struct WTree {
  std::vector<WNode> nodes;
  double children = 0;
};

static float half_area(const float *lo, const float *hi) {
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return (dx * dy + dx * dz) + dy * dz;
}

// greedy collapse of the binary reference tree into W-wide nodes (open the interior child of largest area while slots remain)
static void collapse(const RefBvh &b, int W, bool quantise, WTree *out) {
  struct It { uint32_t node, slot; };
  std::vector<It> todo = {{0u, 0u}};
  out->nodes.assign(1, WNode());
  while (!todo.empty()) {
    const It it = todo.back();
    todo.pop_back();
    std::vector<uint32_t> kids = {it.node + 1, b.nodes[it.node].offset};
    for (;;) {
      int best = -1;
      float ba = -1.f;
      for (size_t k = 0; k < kids.size(); k++) {
        const BvhNode &c = b.nodes[kids[k]];
        if (c.count_axis & 0xffffu) continue;
        const float a = half_area(c.lo, c.hi);
        if ((int)kids.size() + 1 <= W && a > ba) { ba = a; best = (int)k; }
      }
      if (best < 0) break;
      const uint32_t c = kids[best];
      kids[best] = kids.back();
      kids.pop_back();
      kids.push_back(c + 1);
      kids.push_back(b.nodes[c].offset);
    }
    const BvhNode &me = b.nodes[it.node];
    WNode w;
    w.n = (int)kids.size();
    for (int a = 0; a < 3; a++) {
      const float origin = me.lo[a], extent = me.hi[a] - me.lo[a];
      int e = -126;
      if (extent > 0.f) { std::frexp(extent / 255.0f, &e); if (e < -126) e = -126; }
      for (;; e++) {
        const float cell = std::ldexp(1.0f, e);
        bool ok = true;
        for (int k = 0; k < w.n && ok; k++) {
          const BvhNode &c = b.nodes[kids[k]];
          if (!quantise) { w.lo[k][a] = c.lo[a]; w.hi[k][a] = c.hi[a]; continue; }
          int ql = (int)std::floor((c.lo[a] - origin) / cell), qh = (int)std::ceil((c.hi[a] - origin) / cell);
          if (ql < 0) ql = 0;
          if (qh < 0) qh = 0;
          const double o64 = origin, c64 = cell;
          while (ql > 0 && o64 + ql * c64 > (double)c.lo[a]) ql--;
          while (qh <= 255 && o64 + qh * c64 < (double)c.hi[a]) qh++;
          if (ql > 255 || qh > 255 || o64 + ql * c64 > (double)c.lo[a]) { ok = false; break; }
          // the decoded planes, rounded outwards to float (the kernel's margins play this part)
          w.lo[k][a] = std::nextafterf((float)(o64 + ql * c64), -INFINITY);
          w.hi[k][a] = std::nextafterf((float)(o64 + qh * c64), INFINITY);
        }
        if (ok) break;
      }
    }
    for (int k = 0; k < w.n; k++) {
      const BvhNode &c = b.nodes[kids[k]];
      if (c.count_axis & 0xffffu) w.ref[k] = kLeaf | c.offset;
      else {
        w.ref[k] = (uint32_t)out->nodes.size();
        out->nodes.push_back(WNode());
        todo.push_back({kids[k], w.ref[k]});
      }
    }
    out->children += w.n;
    out->nodes[it.slot] = w;
  }
}

struct Stat { double steps = 0, tris = 0, stack = 0; uint64_t bad = 0; };

// events: per ray the sequence of its walk -- 0 = a node step, 1 = a triangle test -- for the scheduling model below
static void walk_all(const WTree &T, const RefBvh &rb, const float *P, const uint32_t *idx, size_t n, const float *o, const float *d, const float *tmax,
                     bool any, bool sorted, const float *rt, const uint32_t *rprim, const uint8_t *rocc, Stat *st,
                     std::vector<std::vector<uint8_t>> *events = nullptr) {
  std::vector<uint32_t> stack;
  for (size_t i = 0; i < n; i++) {
    if (events) events->emplace_back();
    const float ox = o[3 * i], oy = o[3 * i + 1], oz = o[3 * i + 2], dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2];
    const float ix = 1.f / dx, iy = 1.f / dy, iz = 1.f / dz;
    const bool nx = ix < 0.f, ny = iy < 0.f, nz = iz < 0.f;
    float best = INFINITY;
    uint32_t prim = 0xffffffffu;
    bool occ = false;
    stack.clear();
    uint32_t cur = 0;
    size_t max_stack = 0;
    while (cur != kDone) {
      if (!(cur & kLeaf)) {
        const WNode &w = T.nodes[cur];
        st->steps++;
        if (events) events->back().push_back(0);
        const float tfar = std::fmin(best, tmax[i]);
        float key[8];
        int hit[8], nh = 0;
        for (int k = 0; k < w.n; k++) {
          const float tnx = ((nx ? w.hi[k][0] : w.lo[k][0]) - ox) * ix, tfx = ((nx ? w.lo[k][0] : w.hi[k][0]) - ox) * ix;
          const float tny = ((ny ? w.hi[k][1] : w.lo[k][1]) - oy) * iy, tfy = ((ny ? w.lo[k][1] : w.hi[k][1]) - oy) * iy;
          const float tnz = ((nz ? w.hi[k][2] : w.lo[k][2]) - oz) * iz, tfz = ((nz ? w.lo[k][2] : w.hi[k][2]) - oz) * iz;
          const float tn = std::fmax(std::fmax(tnx, tny), std::fmax(tnz, kTMin)), tf = std::fmin(std::fmin(tfx, tfy), std::fmin(tfz, tfar));
          if (tn <= tf * kPad) { key[nh] = tn; hit[nh++] = k; }
        }
        if (nh == 0) { cur = stack.empty() ? kDone : stack.back(); if (!stack.empty()) stack.pop_back(); continue; }
        int nearest = 0;
        for (int h = 1; h < nh; h++) if (key[h] < key[nearest]) nearest = h;
        if (sorted) {  // the others stacked by entry distance, the farthest deepest (what the 8-wide paper's octant order approximates)
          int ord[8], m = 0;
          for (int h = 0; h < nh; h++) if (h != nearest) ord[m++] = h;
          std::sort(ord, ord + m, [&](int a, int c) { return key[a] > key[c]; });
          for (int j = 0; j < m; j++) stack.push_back(w.ref[hit[ord[j]]]);
        } else {  // the product's rule: the others in slot order
          for (int h = nh - 1; h >= 0; h--) if (h != nearest) stack.push_back(w.ref[hit[h]]);
        }
        if (stack.size() > max_stack) max_stack = stack.size();
        cur = w.ref[hit[nearest]];
        continue;
      }
      const uint32_t r = cur & ~kLeaf, id = rb.ref_tri[r];
      st->tris++;
      if (events) events->back().push_back(1);
      const float *a = P + 3 * (size_t)idx[3 * (size_t)id], *b = P + 3 * (size_t)idx[3 * (size_t)id + 1], *c = P + 3 * (size_t)idx[3 * (size_t)id + 2];
      const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
      const float pv[3] = {(dy * e2[2]) - (dz * e2[1]), (dz * e2[0]) - (dx * e2[2]), (dx * e2[1]) - (dy * e2[0])};
      const float det = (e1[0] * pv[0] + e1[1] * pv[1]) + e1[2] * pv[2];
      const float idet = 1.0f / det;
      const float tv[3] = {ox - a[0], oy - a[1], oz - a[2]};
      const float u = ((tv[0] * pv[0] + tv[1] * pv[1]) + tv[2] * pv[2]) * idet;
      const float qv[3] = {(tv[1] * e1[2]) - (tv[2] * e1[1]), (tv[2] * e1[0]) - (tv[0] * e1[2]), (tv[0] * e1[1]) - (tv[1] * e1[0])};
      const float v = ((dx * qv[0] + dy * qv[1]) + dz * qv[2]) * idet;
      const float th = ((e2[0] * qv[0] + e2[1] * qv[1]) + e2[2] * qv[2]) * idet;
      const bool valid = !(std::fabs(det) < 1e-8f) && u >= 0.f && v >= 0.f && u + v <= 1.0f && th > kTMin && th < tmax[i];
      bool stop = false;
      if (valid && any) { occ = true; stop = true; }
      if (valid && !any && (th < best || (th == best && id < prim))) { best = th; prim = id; }
      cur = (stop || stack.empty()) ? kDone : stack.back();
      if (!stop && !stack.empty()) stack.pop_back();
    }
    st->stack += (double)max_stack;
    if (any ? (occ != (rocc[i] != 0)) : (prim != rprim[i] || std::memcmp(&best, &rt[i], 4) != 0)) st->bad++;
  }
}


// ---- scheduling model: how full are a wave's passes? -------------------------------------------------------------------------
// A wave works through `rays` (their event sequences) the way intersect_kernel / render_kernel's traversal loop does -- every lane
// owns one ray, node-step passes (three per scheduling check) run while at least `min_walkers` lanes still walk or nobody waits to be
// served, a leaf pass runs when `min_parked` lanes are parked or the parked lanes are at least half the steppers, a lane whose ray is
// over gets the next one in a service pass -- or, POOLED, keeps `pool` rays in LDS and deals up to 64 rays of ONE state to its lanes per
// pass (the fullest state first; a pass on fewer than `min_pool` rays only when nothing fuller exists).  Counts passes and lanes.
struct Sched { double step_passes = 0, step_lanes = 0, leaf_passes = 0, leaf_lanes = 0, service_passes = 0, service_lanes = 0, rays = 0; };

static void sched_lanes(const std::vector<std::vector<uint8_t>> &ev, size_t first, size_t count, Sched *out) {
  const int kMinWalkers = 36, kMinParked = 16, kSteps = 3;
  struct Lane { long ray = -1; size_t pos = 0; };
  Lane L[64];
  size_t next = first, end = first + count;
  auto done = [&](const Lane &l) { return l.ray < 0 || l.pos >= ev[(size_t)l.ray].size(); };
  for (;;) {
    // service: lanes whose ray is over take the next one
    int finished = 0, walking = 0;
    for (auto &l : L) { if (done(l)) finished++; else walking++; }
    if (walking == 0 && next >= end) break;
    if ((walking < kMinWalkers || walking == 0) && finished > 0 && next < end) {
      int served = 0;
      for (auto &l : L) if (done(l) && next < end) { l.ray = (long)next++; l.pos = 0; served++; out->rays++; }
      out->service_passes++; out->service_lanes += served;
      continue;
    }
    for (int rep = 0; rep < kSteps; rep++) {
      int act = 0;
      for (auto &l : L) if (!done(l) && ev[(size_t)l.ray][l.pos] == 0) { l.pos++; act++; }
      if (act) { out->step_passes++; out->step_lanes += act; }
    }
    int parked = 0, steppers = 0;
    for (auto &l : L) if (!done(l)) { if (ev[(size_t)l.ray][l.pos] == 1) parked++; else steppers++; }
    if (parked && (parked >= kMinParked || parked * 2 >= steppers)) {
      for (auto &l : L) if (!done(l) && ev[(size_t)l.ray][l.pos] == 1) l.pos++;
      out->leaf_passes++; out->leaf_lanes += parked;
    }
  }
}

static void sched_pool(const std::vector<std::vector<uint8_t>> &ev, size_t first, size_t count, int pool, int min_pool, Sched *out) {
  struct Slot { long ray = -1; size_t pos = 0; };
  std::vector<Slot> S((size_t)pool);
  size_t next = first, end = first + count;
  auto done = [&](const Slot &l) { return l.ray < 0 || l.pos >= ev[(size_t)l.ray].size(); };
  for (;;) {
    int ns = 0, nl = 0, nf = 0;
    for (auto &l : S) { if (done(l)) nf++; else if (ev[(size_t)l.ray][l.pos] == 0) ns++; else nl++; }
    const bool refill = nf > 0 && next < end;
    if (ns == 0 && nl == 0 && !refill) break;
    // the fullest pass first; service counts as full when 64 slots can be refilled
    const int cs = std::min(ns, 64), cl = std::min(nl, 64), cf = refill ? std::min<int>(nf, 64) : 0;
    int pick = 0;  // 0 step, 1 leaf, 2 service
    if (cl > cs) pick = 1;
    if (cf > std::max(cs, cl)) pick = 2;
    (void)min_pool;
    if (pick == 0) {
      int k = 0;
      for (auto &l : S) if (k < 64 && !done(l) && ev[(size_t)l.ray][l.pos] == 0) { l.pos++; k++; }
      out->step_passes++; out->step_lanes += k;
    } else if (pick == 1) {
      int k = 0;
      for (auto &l : S) if (k < 64 && !done(l) && ev[(size_t)l.ray][l.pos] == 1) { l.pos++; k++; }
      out->leaf_passes++; out->leaf_lanes += k;
    } else {
      int k = 0;
      for (auto &l : S) if (k < 64 && done(l) && next < end) { l.ray = (long)next++; l.pos = 0; k++; out->rays++; }
      out->service_passes++; out->service_lanes += k;
    }
  }
}

static void sched_report(const char *what, const std::vector<std::vector<uint8_t>> &ev, int W) {
  // waves of a launch each take a contiguous run of 4096 rays (neighbouring rays of the frame: what a wave of the kernel sees)
  const size_t per_wave = 4096;
  const double step_cost = W == 4 ? 88.0 : 88.0 * 1.8, leaf_cost = 70.0, service_cost = 40.0;
  for (int mode = 0; mode < 4; mode++) {
    const int pool = mode == 0 ? 0 : (mode == 1 ? 96 : (mode == 2 ? 128 : 256));
    Sched s;
    for (size_t f = 0; f + per_wave <= ev.size(); f += per_wave) {
      if (pool) sched_pool(ev, f, per_wave, pool, 32, &s); else sched_lanes(ev, f, per_wave, &s);
    }
    if (s.rays == 0) continue;
    // a pooled pass pays for dealing rays to lanes and for reading / writing their state in LDS: + 20 instructions on ~ 100
    const double ovh = pool ? 1.2 : 1.0;
    const double qc = (s.step_passes * step_cost * ovh + s.leaf_passes * leaf_cost * ovh + s.service_passes * service_cost) / s.rays;
    std::printf("  %s, %s: per ray %.3f step passes (%.1f lanes), %.3f leaf passes (%.1f lanes), %.3f refills (%.1f lanes) => %.1f issue quad-cycles per ray (model)\n", what,
                pool ? (pool == 96 ? "pool of  96 rays per wave" : pool == 128 ? "pool of 128 rays per wave" : "pool of 256 rays per wave") : "one ray per lane (today)  ",
                s.step_passes / s.rays, s.step_lanes / std::max(1.0, s.step_passes), s.leaf_passes / s.rays, s.leaf_lanes / std::max(1.0, s.leaf_passes),
                s.service_passes / s.rays, s.service_lanes / std::max(1.0, s.service_passes), qc);
  }
}

int main(int argc, char **argv) {
  if (argc < 2) { std::printf("usage: wide_sim <dir with mesh.bin, closest.bin, shadow.bin>\n"); return 1; }
  const std::string dir = argv[1];
  auto mesh = slurp((dir + "/mesh.bin").c_str());
  const uint32_t n_tris = *(const uint32_t *)mesh.data(), n_verts = *((const uint32_t *)mesh.data() + 1);
  const float *P = (const float *)(mesh.data() + 8);
  const uint32_t *idx = (const uint32_t *)(mesh.data() + 8 + 12 * (size_t)n_verts);
  auto cl = slurp((dir + "/closest.bin").c_str()), sh = slurp((dir + "/shadow.bin").c_str());
  const size_t nc = *(const uint32_t *)cl.data(), ns = *(const uint32_t *)sh.data();
  const float *co = (const float *)(cl.data() + 4), *cd = co + 3 * nc, *ct = cd + 3 * nc, *crt = ct + nc;
  const uint32_t *cprim = (const uint32_t *)(crt + nc);
  const float *so = (const float *)(sh.data() + 4), *sd = so + 3 * ns, *stm = sd + 3 * ns;
  const uint8_t *socc = (const uint8_t *)(stm + ns);
  std::printf("%u triangles, %zu closest-hit rays, %zu shadow rays\n", n_tris, nc, ns);
  Bvh canon;
  build_bvh(P, idx, n_tris, &canon);
  for (int optimise = 0; optimise < 2; optimise++) {
    RefBvh rb;
    single_ref_tree(canon, P, idx, &rb);
    if (optimise) { ReinsertBatchParams prm; prm.passes = 12; reinsert_optimize_batch(&rb, prm); }
    for (int W : {4, 8}) {
      for (int sorted = 0; sorted < 2; sorted++) {
        WTree T;
        collapse(rb, W, true, &T);
        Stat c, s;
        std::vector<std::vector<uint8_t>> evc, evs;
        const bool model = optimise && !sorted;
        walk_all(T, rb, P, idx, nc, co, cd, ct, false, sorted != 0, crt, cprim, nullptr, &c, model ? &evc : nullptr);
        walk_all(T, rb, P, idx, ns, so, sd, stm, true, sorted != 0, nullptr, nullptr, socc, &s, model ? &evs : nullptr);
        const double rays = (double)(nc + ns), steps = (c.steps + s.steps) / rays, tris = (c.tris + s.tris) / rays;
        const int pieces = W == 4 ? 4 : 5;
        std::printf("%s W=%d %s: nodes %8zu (%.2f children per node, %.1f MB) | steps/ray %6.2f tris/ray %5.2f mean deepest stack %.1f | 16-byte L1 accesses/ray %6.1f | hits %s\n",
                    optimise ? "optimised tree" : "binned SAH    ", W, sorted ? "others by distance" : "others in slot order", T.nodes.size(), T.children / T.nodes.size(),
                    T.nodes.size() * (W == 4 ? 64.0 : 80.0) / 1e6, steps, tris, (c.stack + s.stack) / rays, steps * pieces + tris * 3, (c.bad + s.bad) ? "DIFFER" : "equal to the oracle's");
        if (model) {
          sched_report(W == 4 ? "closest-hit rays, W=4" : "closest-hit rays, W=8", evc, W);
          sched_report(W == 4 ? "shadow rays,      W=4" : "shadow rays,      W=8", evs, W);
        }
      }
    }
  }
  return 0;
}
