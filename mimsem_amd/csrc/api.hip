// mimsem_amd/csrc/api.hip -- C-ABI entry points of libmimsem_hip.so (include/mimsem_hip.h):
// context lifetime, HBM layout, scatter-add plans, horizontal-operator dispatch.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include "ctx.hpp"

namespace mimsem {
thread_local std::string g_last_hip_error;
int hip_fail(hipError_t e, const char* what) {
    g_last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
    return MIMSEM_ERR_HIP;
}
}  // namespace mimsem

namespace {

template <class T>
int upload(T** dst, const T* src, size_t count, mimsem_ctx* c) {
    MIMSEM_HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(count, 1)*sizeof(T)));
    c->bytes += (long long)(count*sizeof(T));
    if (count) MIMSEM_HIP_TRY(hipMemcpy(*dst, src, count*sizeof(T), hipMemcpyHostToDevice));
    return MIMSEM_OK;
}

// slot -> list of element-local result positions, ascending element order (deterministic sums)
int build_plan(int nslots, int nEl, int per_el, const std::vector<const int*>& maps, const std::vector<int>& offs,
               int counts, int K, std::vector<int>& plan) {
    plan.assign((size_t)nslots*K, -1);
    std::vector<int> fill(nslots, 0);
    for (int e = 0; e < nEl; e++)
        for (size_t m = 0; m < maps.size(); m++)
            for (int j = 0; j < counts; j++) {
                const int s = maps[m][(size_t)e*counts + j];
                if (s < 0 || s >= nslots) return MIMSEM_ERR_ARG;
                if (fill[s] >= K) return MIMSEM_ERR_UNSUPPORTED;
                plan[(size_t)s*K + fill[s]++] = e*per_el + offs[m] + j;
            }
    return MIMSEM_OK;
}

// Element groups (= workgroups of the fused kernel) grown greedily over shared edge slots, and the group-local
// slot tables: a slot whose every contributor sits in one group is summed in LDS and written once; the others
// ("perimeter") leave one partial sum per group and are finished by k_gather_perim.
struct FusedPlan {
    int ngroups = 0, lmax = 0, nps = 0, npart = 0;
    std::vector<int> perm, fslot, fcnt, pslot, ppart;
    std::vector<unsigned short> lid;
};
int build_fused_plan(int n1, int nEl, int n1e, int G, const int* ix, const int* iy, FusedPlan& P) {
    const int nd = 2*n1e;
    auto slot_of = [&](int e, int j) { return j < n1e ? ix[(size_t)e*n1e + j] : iy[(size_t)e*n1e + j - n1e]; };
    std::vector<int> own((size_t)n1*2, -1), cnt(n1, 0);
    for (int e = 0; e < nEl; e++) for (int j = 0; j < nd; j++) {
        const int s = slot_of(e, j);
        if (cnt[s] >= 2) return MIMSEM_ERR_UNSUPPORTED;
        own[(size_t)s*2 + cnt[s]++] = e;
    }
    P.ngroups = (nEl + G - 1)/G; P.lmax = G*nd;
    P.perm.assign((size_t)P.ngroups*G, -1);
    std::vector<char> assigned(nEl, 0);
    std::vector<int> score(nEl, 0), touched;
    int next_seed = 0;
    for (int g = 0; g < P.ngroups; g++) {
        for (int t : touched) score[t] = 0;
        touched.clear();
        for (int k = 0; k < G; k++) {
            int best = -1;
            for (int t : touched) if (!assigned[t] && (best < 0 || score[t] > score[best] || (score[t] == score[best] && t < best))) best = t;
            if (best < 0) { while (next_seed < nEl && assigned[next_seed]) next_seed++; if (next_seed >= nEl) break; best = next_seed; }
            assigned[best] = 1; P.perm[(size_t)g*G + k] = best;
            for (int j = 0; j < nd; j++) {
                const int s = slot_of(best, j);
                for (int w = 0; w < 2; w++) { const int o = own[(size_t)s*2 + w]; if (o >= 0 && o != best && !assigned[o]) { if (!score[o]) touched.push_back(o); score[o]++; } }
            }
        }
    }
    // per group: its distinct slots in ASCENDING slot order (coalesced write-out); for each the one or two
    // element-local result positions (k*nd + j inside the group) that feed it, in element order
    P.lid.assign((size_t)P.ngroups*P.lmax*2, 0xFFFF);
    P.fslot.assign((size_t)P.ngroups*P.lmax, 0); P.fcnt.assign(P.ngroups, 0);
    std::vector<int> part((size_t)n1*2, -1);
    std::vector<int> ingroup(n1, 0);
    for (int g = 0; g < P.ngroups; g++) {
        std::vector<std::pair<int, int>> contrib;                      // (slot, position in group)
        for (int k = 0; k < G; k++) { const int e = P.perm[(size_t)g*G + k]; if (e < 0) continue;
            for (int j = 0; j < nd; j++) { const int s = slot_of(e, j); ingroup[s]++; contrib.push_back({s, k*nd + j}); } }
        std::sort(contrib.begin(), contrib.end());
        int nl = 0;
        for (size_t i = 0; i < contrib.size(); ) {
            const int s = contrib[i].first;
            size_t j = i; while (j < contrib.size() && contrib[j].first == s) j++;
            P.lid[((size_t)g*P.lmax + nl)*2 + 0] = (unsigned short)contrib[i].second;
            if (j - i > 1) P.lid[((size_t)g*P.lmax + nl)*2 + 1] = (unsigned short)contrib[i + 1].second;
            if (ingroup[s] == cnt[s]) P.fslot[(size_t)g*P.lmax + nl] = s;
            else { const int pi = P.npart++; P.fslot[(size_t)g*P.lmax + nl] = -(pi + 1);
                   if (part[(size_t)s*2] < 0) part[(size_t)s*2] = pi; else part[(size_t)s*2 + 1] = pi; }
            nl++; i = j;
        }
        P.fcnt[g] = nl;
        for (auto& c2 : contrib) ingroup[c2.first] = 0;
    }
    for (int s = 0; s < n1; s++) if (part[(size_t)s*2] >= 0) { P.pslot.push_back(s); P.ppart.push_back(part[(size_t)s*2]); P.ppart.push_back(part[(size_t)s*2 + 1]); }
    P.nps = (int)P.pslot.size();
    return MIMSEM_OK;
}

// Wave-groups of the wave-level fused kernel (k_apply_wave): G = 64/LPE elements grown over shared edge slots so that they form a
// compact patch (2 x 2 at p <= 3: the candidate sharing the most slots with the group wins, ties go to a neighbour of the SEED, which
// turns the third pick perpendicular to the second and lets the fourth close the square), and per group three lane tables.  The
// element kernels are bound by the number of vector-memory instructions a wave issues (TA/TD busy 85 %, profiles/r02_umat_pmc.txt),
// so both tables are built around 16-BYTE accesses:
//   load pairs : the group's distinct slots covered by aligned pairs {2k, 2k+1} (the reference numbers an element's x- and y-edge
//                DoFs alternately, so a pair is usually two wanted values): ONE dwordx4 load per level brings every DoF of the
//                group; each loaded value is staged to the one or two elements that use it.
//   store pairs: a pair of slots that are both COMPLETE in the group (every contributor inside it) is summed in LDS and written
//                once, straight into y, as one 16-byte store; so is a complete slot whose pair partner is a perimeter slot of the same
//                group (the partner's half of the store carries 0 and is overwritten by the perimeter pass later in the stream);
//                every other slot of the group (its perimeter, and complete slots whose partner is outside the group) leaves a
//                partial sum in a densely packed row of the workspace -- again in pairs -- that k_wave_perim finishes.
//                MIMSEM_WAVE_SINGLES=1: no mixed pairs, unpaired complete slots in a second, 8-byte store round instead.
constexpr int WMP = 16;     // entries of a side (its slots, padded): 8 pairs x 8 levels x 2 parts = two 64-lane rounds of the finishing phase
constexpr int WNS = 8;      // sides a wave-group can take part in
struct WavePlan {
    int ngroups = 0, nps = 0, npart = 0, npwritten = 0, nsing = 0, ndirect = 0, nbgroups = 0, nbrec = 0, nsides = 0;
    bool fin_ok = true;
    std::vector<int> perm, pslot, ppart, node, sslot;
    std::vector<int4> lane, plan;
    std::vector<int2> sing;
    std::vector<int4> fin;
    // round 5, tile mode: four consecutive wave-groups = one workgroup = a tile; the partial sums of the slots two groups of a tile share
    // meet in the workgroup's LDS instead of the workspace (tfin: [ntiles][WTF] {slot, LDS position of part A, of part B, 0}, -1 padded)
    int ntiles = 0, tpmax = 0, ninner = 0;
    std::vector<int4> tfin;
};
constexpr int WTF = MIMSEM_WTF;      // finishing entries a tile can hold (4 x 4 elements at p = 3: 24 inner slots)
constexpr int WTP = MIMSEM_WTP;      // doubles of a tile's LDS row (pairs of partial sums)
// Round 3: the partial sums are laid out per SIDE -- the (<= WMP) perimeter slots two wave-groups share, in ascending slot order; row
// of a level = [side][part A (lower group) | part B][WMP] -- so that whichever group reaches a side second can finish its slots inside
// the same launch (elem_wave.inc, finishing phase): no second kernel, no second trip of the partial sums through HBM.  A complete
// slot that cannot be written in a 16-byte pair (its partner lies outside the group) rides along as an extra entry of one of the
// group's sides (the other part of that entry is written as 0).  The perimeter records (k_wave_perim: split applies, MIMSEM_WAVE_FIN=0)
// address the same layout.
int build_wave_plan(int order, int n1, int nEl, int n1e, int n0e, int G, const int* ix, const int* iy, const int* i0, bool singles, bool mixed,
                    const char* marked /* [n1] halo slots or null */, WavePlan& P, bool tile = false) {
    const int nd = 2*n1e, lpe = 64/G, mp1 = order + 1;
    const int RS = mp1 + (mp1 & 1), XT = n1e + mp1*RS, sxe0 = XT + 2*(lpe - n1e), SXE = sxe0 + (sxe0 & 1);     // as k_apply_wave
    const unsigned NACC = (unsigned)(G*nd), ZERO = NACC + 128, DUMPX = (unsigned)XT;
    if (G*SXE > 0xFFFE || n1 < 2) return MIMSEM_ERR_UNSUPPORTED;
    auto slot_of = [&](int e, int j) { return j < n1e ? ix[(size_t)e*n1e + j] : iy[(size_t)e*n1e + j - n1e]; };
    auto xpos = [&](int k, int j) { return (unsigned)(k*SXE + (j < n1e ? j : n1e + ((j - n1e)/order)*RS + (j - n1e)%order)); };
    std::vector<int> own((size_t)n1*2, -1), cnt(n1, 0);
    for (int e = 0; e < nEl; e++) for (int j = 0; j < nd; j++) {
        const int s = slot_of(e, j);
        if (s < 0 || s >= n1) return MIMSEM_ERR_ARG;
        if (cnt[s] >= 2) return MIMSEM_ERR_UNSUPPORTED;
        own[(size_t)s*2 + cnt[s]++] = e;
    }
    P.ngroups = (nEl + G - 1)/G;
    P.perm.assign((size_t)P.ngroups*G, -1);
    std::vector<char> assigned(nEl, 0), seedn(nEl, 0);
    std::vector<int> score(nEl, 0), touched;
    int next_seed = 0;
    for (int g = 0; g < P.ngroups; g++) {
        for (int t : touched) { score[t] = 0; seedn[t] = 0; }
        touched.clear();
        for (int k = 0; k < G; k++) {
            int best = -1;
            auto better = [&](int t, int b) {
                if (score[t] != score[b]) return score[t] > score[b];
                if (seedn[t] != seedn[b]) return seedn[t] > seedn[b];
                return t < b; };
            for (int t : touched) if (!assigned[t] && (best < 0 || better(t, best))) best = t;
            if (best < 0) { while (next_seed < nEl && assigned[next_seed]) next_seed++; if (next_seed >= nEl) break; best = next_seed; }
            assigned[best] = 1; P.perm[(size_t)g*G + k] = best;
            for (int j = 0; j < nd; j++) {
                const int s = slot_of(best, j);
                for (int w = 0; w < 2; w++) {
                    const int o = own[(size_t)s*2 + w];
                    if (o >= 0 && o != best && !assigned[o]) { if (!score[o]) touched.push_back(o); score[o]++; if (k == 0) seedn[o] = 1; }
                }
            }
        }
    }
    // interior / boundary split (mimsem_ctx_set_halo_slots): groups that touch a slot taking part in a halo exchange come FIRST, so
    // that "the boundary part" of an apply is a prefix of the launch (and a prefix of the perimeter records, below)
    P.nbgroups = 0;
    if (marked) {
        std::vector<int> bnd, inn;
        for (int g = 0; g < P.ngroups; g++) {
            bool b = false;
            for (int k = 0; k < G && !b; k++) { const int e = P.perm[(size_t)g*G + k]; if (e < 0) continue;
                for (int j = 0; j < nd && !b; j++) b = marked[slot_of(e, j)] != 0; }
            (b ? bnd : inn).push_back(g);
        }
        std::vector<int> perm2; perm2.reserve(P.perm.size());
        for (int g : bnd) perm2.insert(perm2.end(), P.perm.begin() + (size_t)g*G, P.perm.begin() + (size_t)(g + 1)*G);
        for (int g : inn) perm2.insert(perm2.end(), P.perm.begin() + (size_t)g*G, P.perm.begin() + (size_t)(g + 1)*G);
        P.perm.swap(perm2); P.nbgroups = (int)bnd.size();
    }
    // ---- tile mode (round 5): wave-groups in fours, each four = the wavefronts of one workgroup.  Grown over the group adjacency (shared
    // perimeter slots) with one step of look-ahead so that the fours come out as 2 x 2 squares of groups where the mesh allows (a square
    // keeps 4 sides inside, a line of four only 3); a four that cannot be filled is padded with empty groups.
    tile = tile && mixed && !singles && !marked;
    if (tile) {
        std::vector<int> g0(nEl, -1);
        for (int g = 0; g < P.ngroups; g++) for (int k = 0; k < G; k++) { const int e = P.perm[(size_t)g*G + k]; if (e >= 0) g0[e] = g; }
        std::vector<std::vector<std::pair<int, int>>> adj(P.ngroups);       // (neighbour group, shared slots)
        {
            std::vector<std::pair<std::pair<int, int>, int>> pr;
            for (int sl = 0; sl < n1; sl++) if (cnt[sl] == 2) {
                const int a = g0[own[(size_t)sl*2]], b = g0[own[(size_t)sl*2 + 1]];
                if (a != b) { pr.push_back({{a, b}, 1}); pr.push_back({{b, a}, 1}); }
            }
            std::sort(pr.begin(), pr.end());
            for (size_t i = 0; i < pr.size();) { size_t j = i; while (j < pr.size() && pr[j].first == pr[i].first) j++;
                adj[pr[i].first.first].push_back({pr[i].first.second, (int)(j - i)}); i = j; }
        }
        std::vector<char> done(P.ngroups, 0);
        auto shared_with = [&](const std::vector<int>& t, int h) { int w = 0; for (int m : t) for (auto& e : adj[m]) if (e.first == h) w += e.second; return w; };
        std::vector<int> perm2;
        for (int seed = 0; seed < P.ngroups; seed++) {
            if (done[seed]) continue;
            std::vector<int> t{seed}; done[seed] = 1;
            while ((int)t.size() < 4) {
                int best = -1, bw = 0, bl = -1;
                for (int m : t) for (auto& e : adj[m]) {
                    const int h = e.first;
                    if (done[h]) continue;
                    const int w = shared_with(t, h);
                    int look = 0;                                       // the most a fourth / next group could share with the tile grown by h
                    if ((int)t.size() < 3) { std::vector<int> t2(t); t2.push_back(h);
                        for (int m2 : t2) for (auto& e2 : adj[m2]) if (!done[e2.first] && e2.first != h) look = std::max(look, shared_with(t2, e2.first)); }
                    if (w > bw || (w == bw && look > bl) || (w == bw && look == bl && h < best)) { best = h; bw = w; bl = look; }
                }
                if (best < 0) break;
                t.push_back(best); done[best] = 1;
            }
            for (int m : t) perm2.insert(perm2.end(), P.perm.begin() + (size_t)m*G, P.perm.begin() + (size_t)(m + 1)*G);
            for (int k = (int)t.size(); k < 4; k++) perm2.insert(perm2.end(), (size_t)G, -1);
            P.ntiles++;
        }
        P.perm.swap(perm2); P.ngroups = 4*P.ntiles;
    }
    std::vector<int> grp(nEl, -1);
    for (int g = 0; g < P.ngroups; g++) for (int k = 0; k < G; k++) { const int e = P.perm[(size_t)g*G + k]; if (e >= 0) grp[e] = g; }
    P.lane.assign((size_t)P.ngroups*64, int4{0, 0, 0, 0});
    P.plan.resize((size_t)P.ngroups*64);
    P.node.resize((size_t)P.ngroups*64);
    if (singles) P.sing.assign((size_t)P.ngroups*64, int2{-1, (int)(ZERO | (ZERO << 16))});
    struct Use { int nx = 0; unsigned x[2] = {0, 0}; int na = 0; unsigned a[2] = {0, 0}; };      // staging / result positions of a slot in the group
    std::vector<Use> use(n1);
    std::vector<std::vector<int4>> entries(P.ngroups);
    std::vector<std::vector<int>> extras(P.ngroups), gsides(P.ngroups), routed_of(P.ngroups);
    // the slots of group g with their positions in its staging / result strips (filled, used, cleared per group)
    auto mark_group = [&](int g, std::vector<int>& slots) {
        slots.clear();
        for (int k = 0; k < G; k++) { const int e = P.perm[(size_t)g*G + k]; if (e < 0) continue;
            for (int j = 0; j < nd; j++) {
                const int s = slot_of(e, j); Use& u = use[s];
                if (!u.nx) slots.push_back(s);
                u.x[u.nx++] = xpos(k, j); u.a[u.na++] = (unsigned)(k*nd + j);
            } }
        std::sort(slots.begin(), slots.end());
    };
    auto in_group = [&](int s) { return s >= 0 && s < n1 && use[s].nx > 0; };
    auto complete = [&](int s) { return in_group(s) && use[s].nx == cnt[s]; };
    auto xpack = [&](int s) { unsigned a0 = DUMPX, a1 = DUMPX; if (in_group(s)) { a0 = use[s].x[0]; if (use[s].nx > 1) a1 = use[s].x[1]; } return (int)(a0 | (a1 << 16)); };
    auto apack = [&](int s) { unsigned a0 = ZERO, a1 = ZERO; if (in_group(s)) { a0 = use[s].a[0]; if (use[s].na > 1) a1 = use[s].a[1]; } return (int)(a0 | (a1 << 16)); };
    // sides: (lower group, upper group) -> the perimeter slots they share
    std::vector<std::pair<std::pair<int, int>, int>> shared;               // ((gA, gB), slot): every perimeter slot is reported by both groups
    std::vector<int> slots;
    for (int g = 0; g < P.ngroups; g++) {
        mark_group(g, slots);
        // lane table: element + node slot of every lane, load pair of the first lanes
        for (int l = 0; l < 64; l++) {
            const int pe = P.perm[(size_t)g*G + l/lpe], e = pe >= 0 ? pe : 0, q = l%lpe;
            P.lane[(size_t)g*64 + l] = int4{e, 0, (int)(DUMPX | (DUMPX << 16)), (int)(DUMPX | (DUMPX << 16))};
            P.node[(size_t)g*64 + l] = i0[(size_t)e*n0e + std::min(q, n0e - 1)];
        }
        int nload = 0; int last = -1;
        for (int s : slots) {
            if (s <= last) continue;                                  // covered by the previous pair
            int b = s & ~1; if (b + 1 >= n1) b = n1 - 2;
            if (nload >= 64) return MIMSEM_ERR_UNSUPPORTED;
            int4& L = P.lane[(size_t)g*64 + nload++];
            L.y = b; L.z = xpack(b); L.w = xpack(b + 1);
            last = b + 1;
        }
        for (int l = nload; l < 64 && nload; l++) P.lane[(size_t)g*64 + l].y = P.lane[(size_t)g*64].y;   // idle loader lanes re-read the first pair
        // store pairs written straight into y
        std::vector<int> routed;
        for (size_t i = 0; i < slots.size(); i++) {
            const int s = slots[i];
            if (!(s & 1) && s + 1 < n1 && complete(s) && complete(s + 1)) { entries[g].push_back(int4{s, apack(s), apack(s + 1), 0}); P.ndirect += 2; i++; continue; }
            if (mixed && !(s & 1) && in_group(s + 1) && complete(s) != complete(s + 1) && !singles) {
                // MIXED pair (MIMSEM_WAVE_MIXED=1: round 2's form, perimeter pass only), both slots in the group: the complete one is
                // written in a full 16-byte store whose other half carries 0 (accumulate form: adds 0).  That half belongs to a
                // perimeter slot NO group writes directly -- its value comes from k_wave_perim, strictly later in the stream --, so the
                // store clobbers nothing.  Inside ONE launch nothing orders that 0 -- dirty in the storing XCD's L2 until the kernel
                // ends -- before the value a finishing wave on another XCD writes: with the finishing phase such a complete slot rides
                // through the side as an extra entry instead.  (A complete slot whose partner is not in the group at all stays
                // routed: the partner may be written directly by its own group at the same time.)
                const int zz = (int)(ZERO | (ZERO << 16));
                entries[g].push_back(int4{s, complete(s) ? apack(s) : zz, complete(s + 1) ? apack(s + 1) : zz, 0});
                P.ndirect += 1;
                routed.push_back(complete(s) ? s + 1 : s);
                i++; continue;
            }
            routed.push_back(s);
        }
        if (singles) {                                                 // complete slots without a complete partner: 8-byte stores
            std::vector<int> keep; int ns = 0;
            for (int s : routed) {
                if (complete(s)) { if (ns >= 64) return MIMSEM_ERR_UNSUPPORTED; P.sing[(size_t)g*64 + ns++] = int2{s, apack(s)}; }
                else keep.push_back(s);
            }
            routed.swap(keep); P.nsing += ns; P.ndirect += ns;
        }
        routed_of[g] = routed;
        for (int s : routed) {
            if (mixed) break;                                          // dense layout below: no sides
            if (complete(s)) { extras[g].push_back(s); continue; }
            int go = -1;
            for (int w = 0; w < 2; w++) { const int o = own[(size_t)s*2 + w]; if (o >= 0 && grp[o] != g) go = grp[o]; }
            if (go < 0) return MIMSEM_ERR_STATE;                       // (cannot happen: an incomplete slot has a contributor elsewhere)
            shared.push_back({{std::min(g, go), std::max(g, go)}, s});
        }
        for (int s : slots) use[s] = Use();
    }
    std::sort(shared.begin(), shared.end());
    shared.erase(std::unique(shared.begin(), shared.end()), shared.end());
    struct Side { int ga, gb; std::vector<int> slots; };
    std::vector<Side> sides;
    for (size_t i = 0; i < shared.size(); i++) {
        const bool fresh = sides.empty() || sides.back().ga != shared[i].first.first || sides.back().gb != shared[i].first.second ||
                           (int)sides.back().slots.size() >= WMP;
        if (fresh) { sides.push_back(Side{shared[i].first.first, shared[i].first.second, {}});
                     gsides[shared[i].first.first].push_back((int)sides.size() - 1); gsides[shared[i].first.second].push_back((int)sides.size() - 1); }
        sides.back().slots.push_back(shared[i].second);
    }
    for (int g = 0; g < P.ngroups; g++)
        for (int s : extras[g]) {
            int sd = -1;
            for (int t : gsides[g]) if ((int)sides[t].slots.size() < WMP) { sd = t; break; }
            if (sd < 0) { sides.push_back(Side{g, -1, {}}); sd = (int)sides.size() - 1; gsides[g].push_back(sd); }     // a side of the group's own
            sides[sd].slots.push_back(s);
        }
    P.nsides = (int)sides.size();
    P.npart = P.nsides*2*WMP;
    P.sslot.assign((size_t)P.nsides*WMP, -1);
    P.fin.assign((size_t)P.ngroups*WNS, int4{-1, 0, 0, 0});
    std::vector<int> part((size_t)n1*2, -1);
    for (int t = 0; t < P.nsides; t++)
        for (size_t j = 0; j < sides[t].slots.size(); j++) P.sslot[(size_t)t*WMP + j] = sides[t].slots[j];
    if (mixed) {
        // round 2's dense layout (the default: the perimeter pass finishes every slot): the routed slots of a group in pairs, one after
        // the other in the row -- 26 values in three cache lines per group and level
        P.npart = 0;
        std::vector<int> tpos(std::max(P.ntiles, 1), 0), tpart;             // doubles used in a tile's LDS row; LDS positions of an inner slot's two parts
        if (tile) { tpart.assign((size_t)n1*2, -1); P.tfin.assign((size_t)P.ntiles*WTF, int4{-1, 0, 0, 0}); }
        std::vector<int> tcount(std::max(P.ntiles, 1), 0);
        for (int g = 0; g < P.ngroups; g++) {
            if (routed_of[g].empty()) continue;
            mark_group(g, slots);
            std::vector<int> routed, inner;
            for (int s : routed_of[g]) {
                bool in = false;
                if (tile && cnt[s] == 2) { const int a = grp[own[(size_t)s*2]], b = grp[own[(size_t)s*2 + 1]]; in = a != b && a/4 == b/4; }
                (in ? inner : routed).push_back(s);
            }
            for (size_t i = 0; i < routed.size(); i += 2) {
                const int s0 = routed[i], s1 = i + 1 < routed.size() ? routed[i + 1] : -1;
                const int pi = P.npart; P.npart += 2; P.npwritten += 2;
                auto reg = [&](int s, int idx) { if (s < 0) return; if (part[(size_t)s*2] < 0) part[(size_t)s*2] = idx; else part[(size_t)s*2 + 1] = idx; };
                reg(s0, pi); reg(s1, pi + 1);
                entries[g].push_back(int4{-(pi + 2), apack(s0), apack(s1), 0});
            }
            // the slots this group shares with another group of its tile: pairs of partial sums into the tile's LDS row (w = position + 1;
            // x = INT_MIN: the lane's global store goes to the dump tail)
            const int T = g/4;
            for (size_t i = 0; i < inner.size(); i += 2) {
                const int s0 = inner[i], s1 = i + 1 < inner.size() ? inner[i + 1] : -1;
                const int tp = tpos[T]; tpos[T] += 2;
                if (tpos[T] > WTP) return MIMSEM_ERR_UNSUPPORTED;
                auto reg = [&](int s, int pos) {
                    if (s < 0) return;
                    if (tpart[(size_t)s*2] < 0) { tpart[(size_t)s*2] = pos; return; }
                    if (tcount[T] >= WTF) return;                         // (checked below)
                    P.tfin[(size_t)T*WTF + tcount[T]] = int4{s, tpart[(size_t)s*2], pos, 0};       // part A = the lower group, as in the perimeter pass
                    tcount[T]++; P.ninner++;
                };
                reg(s0, tp); reg(s1, tp + 1);
                entries[g].push_back(int4{INT_MIN, apack(s0), apack(s1), tp + 1});
            }
            if (entries[g].size() > 64) return MIMSEM_ERR_UNSUPPORTED;
            for (int s : slots) use[s] = Use();
        }
        if (tile) {
            for (int T = 0; T < P.ntiles; T++) { if (tcount[T] >= WTF) return MIMSEM_ERR_UNSUPPORTED; P.tpmax = std::max(P.tpmax, tpos[T]); }
            for (int s = 0; s < n1; s++) if (tpart[(size_t)s*2] >= 0) {       // every inner slot must have found its second part
                bool found = false;
                for (int k = 0; k < WTF && !found; k++) { const int4& r = P.tfin[(size_t)(grp[own[(size_t)s*2]]/4)*WTF + k]; found = r.x == s; }
                if (!found) return MIMSEM_ERR_STATE;
            }
        }
    }
    for (int g = 0; g < P.ngroups && !mixed; g++) {
        if ((int)gsides[g].size() > WNS) P.fin_ok = false;
        if (gsides[g].empty()) continue;
        mark_group(g, slots);
        for (size_t k = 0; k < gsides[g].size(); k++) {
            const int t = gsides[g][k]; const Side& S = sides[t];
            const int base = t*2*WMP + (g == S.ga ? 0 : WMP);
            if (k < (size_t)WNS) P.fin[(size_t)g*WNS + k] = int4{t, S.gb < 0 ? 1 : 2, (int)S.slots.size(), 0};
            // the whole 128-byte part, padding included, by ONE store instruction (8 lanes x 16 bytes): a full line travels to the
            // uncached row as one write -- partial-line writes are read-modify-writes at the memory side (measured: 23 scattered
            // 16-byte pairs per group and level made the launch 3x slower)
            size_t need = 0;                                              // (groups with many sides: no room for the padding pairs)
            for (size_t k2 = k; k2 < gsides[g].size(); k2++) need += (sides[gsides[g][k2]].slots.size() + 1)/2;
            const bool whole = entries[g].size() + (gsides[g].size() - k)*(WMP/2) <= 64 || entries[g].size() + need + (WMP/2 - (S.slots.size() + 1)/2) > 64 ? 
                               entries[g].size() + (gsides[g].size() - k)*(WMP/2) <= 64 : false;
            for (size_t j = 0; j < (whole ? (size_t)WMP : S.slots.size()); j += 2) {
                const int s0 = j < S.slots.size() ? S.slots[j] : -1, s1 = j + 1 < S.slots.size() ? S.slots[j + 1] : -1;
                entries[g].push_back(int4{-(base + (int)j + 2), apack(s0), apack(s1), 0});     // a slot the group has no share of: 0
                if (s0 >= 0) P.npwritten += 2;
            }
            for (size_t j = 0; j < S.slots.size(); j++)
                if (in_group(S.slots[j])) { const int s = S.slots[j], idx = base + (int)j;
                    if (part[(size_t)s*2] < 0) part[(size_t)s*2] = idx;
                    else if (idx < part[(size_t)s*2]) { part[(size_t)s*2 + 1] = part[(size_t)s*2]; part[(size_t)s*2] = idx; }
                    else part[(size_t)s*2 + 1] = idx; }               // part A first: the perimeter pass and the finishing phase add A + B
        }
        if (entries[g].size() > 64) return MIMSEM_ERR_UNSUPPORTED;
        for (int s : slots) use[s] = Use();
    }
    // unused lanes REPEAT one of the group's own entries (same address, same value: the duplicates merge in the store instruction and
    // touch no extra cache line); a group without any entry (padding only) stores into the dump tail of the partial-sum row
    for (int g = 0; g < P.ngroups; g++) {
        const int ne = (int)entries[g].size();
        for (int t = 0; t < 64; t++)
            P.plan[(size_t)g*64 + t] = ne ? entries[g][t%ne]
                                          : int4{-(P.npart + 2*t + 2), (int)(ZERO | (ZERO << 16)), (int)(ZERO | (ZERO << 16)), 0};
    }
    // perimeter records: those of marked (halo) slots first; inside each segment in SLOT order (coalesced y accesses; ordering by
    // the first partial sum instead was measured: the pass went from 30 to 38 us on the 829 440-unit launch)
    P.nbrec = 0;
    for (int pass = 0; pass < 2; pass++) {
        std::vector<std::pair<int, int>> seg;                         // (first partial or INT_MAX, slot)
        for (int s = 0; s < n1; s++) {
            const bool m = marked && marked[s];
            if ((pass == 0) != m) continue;
            if (cnt[s] == 0) P.fin_ok = false;                        // a slot no element touches: only the perimeter pass zeroes it
            if (part[(size_t)s*2] >= 0 || cnt[s] == 0) seg.push_back({s, s});
        }
        std::sort(seg.begin(), seg.end());
        for (auto& e : seg) { const int s = e.second; P.pslot.push_back(s); P.ppart.push_back(part[(size_t)s*2]); P.ppart.push_back(part[(size_t)s*2 + 1]); }
        if (pass == 0) P.nbrec = (int)seg.size();
    }
    P.nps = (int)P.pslot.size();
    if (singles || mixed) P.fin_ok = false;
    return MIMSEM_OK;
}

// levels per work item of the element kernel: keep >= ~6 workgroups per CU in flight, otherwise amortise as much as possible
int level_chunk(const mimsem_ctx* c, int nlev) {
    const ElemSizes& es = c->es;
    const int epb = 256/(es.mp12 <= 4 ? 4 : (es.mp12 <= 16 ? 16 : (es.mp12 <= 32 ? 32 : 64)));
    const long long blocks1 = ((long long)c->nEl + epb - 1)/epb;          // workgroups per single level
    long long lch = (blocks1*nlev)/(256*6);
    if (lch < 1) lch = 1;
    if (lch > nlev) lch = nlev;
    if (lch > 8) lch = 8;
    if (c->lch_override > 0) lch = std::min(c->lch_override, nlev);
    return (int)std::max(1LL, lch);
}

// levels per work item of the wave-level kernel: its compile-time bound WLC = 8 whenever the call has that many levels.  A wavefront's
// fixed costs (kernel arguments, tables, metric, the dispatch of the wave itself: ~25 cycles per wave and XCD) are what the
// 103 680-unit launch is made of (scripts/wave_size_sweep.py, s_memtime stamps of scripts/stamp_wave.sh), and the kernel computes
// all WLC levels of a work item whether the chunk has them or not -- so: as many levels per wave as there are.
int wave_level_chunk(const mimsem_ctx* c, int nlev) {
    if (c->wave_lch > 0) return std::max(1, std::min(std::min(c->wave_lch, 8), nlev));
    return std::max(1, std::min(8, nlev));
}

// chunks of 8 levels one wavefront works through (tables and metric loaded once, the level pipeline running on across the chunk
// boundary): as many as leave ~2 wavefronts per SIMD -- fewer, longer-lived waves cost less dispatch and set-up, which is what the
// cache-resident launches are made of
int wave_chunks_per_item(const mimsem_ctx* c, int nlev, int lch, int ngroups) {
    const int nch = (nlev + lch - 1)/lch;
    if (lch != 8 || nch <= 1) return 1;
    if (c->wave_cpp > 0) return std::min(c->wave_cpp, nch);
    int nparts = std::min(nch, std::max(1, (1536 + ngroups - 1)/std::max(ngroups, 1)));
    // EQUAL parts where the chunk count allows: the launch ends with its longest work item (p = 4 box, 8 chunks: 3 parts of 3 + 3 + 2
    // chunks took 17.2 us, 4 parts of 2 take 14.0 us -- round 3)
    for (int np = nparts; np <= nch; np++) if (nch%np == 0) { nparts = np; break; }
    return (nch + nparts - 1)/nparts;
}

int op_spaces(int op, int* in, int* cf, int* out) {
    switch (op) {
    case MIMSEM_OP_UMAT: case MIMSEM_OP_UTMAT:   *in = 1; *cf = -1; *out = 1; return 0;
    case MIMSEM_OP_UHMAT: case MIMSEM_OP_UTMAT_H: *in = 1; *cf = 2; *out = 1; return 0;
    case MIMSEM_OP_ROTMAT:  *in = 1; *cf = 0; *out = 1; return 0;
    case MIMSEM_OP_WMAT: case MIMSEM_OP_WMATINV:  *in = 2; *cf = -1; *out = 2; return 0;
    case MIMSEM_OP_WHMAT: case MIMSEM_OP_WHMATINV: *in = 2; *cf = 2; *out = 2; return 0;
    case MIMSEM_OP_PMAT:    *in = 0; *cf = -1; *out = 0; return 0;
    case MIMSEM_OP_PHMAT:   *in = 0; *cf = 2; *out = 0; return 0;
    case MIMSEM_OP_WTQUMAT: case MIMSEM_OP_WTQDUDZ: *in = 1; *cf = 1; *out = 2; return 0;
    case MIMSEM_OP_UTQWMAT: *in = 2; *cf = 1; *out = 1; return 0;
    case MIMSEM_OP_WTQ: *in = 3; *cf = -1; *out = 2; return 0;
    case MIMSEM_OP_PTQ: *in = 3; *cf = -1; *out = 0; return 0;
    case MIMSEM_OP_UTQ: *in = 3; *cf = -1; *out = 1; return 0;
    case MIMSEM_OP_PHMAT_UP:  *in = 0; *cf = 2; *out = 0; return 0;
    case MIMSEM_OP_ROTMAT_UP: *in = 1; *cf = 0; *out = 1; return 0;
    case MIMSEM_OP_UMAT_UP:   *in = 1; *cf = 1; *out = 1; return 0;
    case MIMSEM_OP_UHMAT_UP: case MIMSEM_OP_UVEC_HU_UP: case MIMSEM_OP_UMAT_RAY: *in = 1; *cf = 2; *out = 1; return 0;
    }
    return 1;
}

}  // namespace

// Workspaces only ever grow, and an outgrown buffer is RETIRED, not freed: a hipGraph captured earlier (Krylov / Richardson steps)
// has the old address baked into its kernel arguments and must keep working on it; retired buffers go with the context.
bool mimsem_ctx::is_capturing() const {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) return false;
    return st != hipStreamCaptureStatusNone;
}
int mimsem_ctx::ensure_ye(long long doubles) {
    if (doubles <= ye_doubles) return MIMSEM_OK;
    if (is_capturing()) return MIMSEM_ERR_STATE;
    if (d_ye) { retired.push_back(d_ye); d_ye = nullptr; ye_doubles = 0; }
    MIMSEM_HIP_TRY(hipMalloc((void**)&d_ye, (size_t)doubles*sizeof(double)));
    ye_doubles = doubles; bytes += doubles*8;
    return MIMSEM_OK;
}
int mimsem_ctx::ensure_cheb(long long doubles) {
    if (doubles <= cheb_doubles) return MIMSEM_OK;
    if (is_capturing()) return MIMSEM_ERR_STATE;
    if (d_cheb) { retired.push_back(d_cheb); d_cheb = nullptr; cheb_doubles = 0; }
    MIMSEM_HIP_TRY(hipMalloc((void**)&d_cheb, (size_t)doubles*sizeof(double)));
    cheb_doubles = doubles; bytes += doubles*8;
    return MIMSEM_OK;
}
int mimsem_ctx::ensure_wpart(long long doubles) {
    if (doubles <= wpart_doubles) return MIMSEM_OK;
    if (is_capturing()) return MIMSEM_ERR_STATE;
    if (d_wpart) { retired.push_back(d_wpart); d_wpart = nullptr; wpart_doubles = 0; }
    // UNCACHED device memory: a plain store is acknowledged by the memory side, not by the storing XCD's L2 (MIMSEM_WPART_MEM=finegrained |
    // plain select other kinds for experiments; `plain` is NOT coherent across XCDs inside one launch)
    if (w_partmem == 2) MIMSEM_HIP_TRY(hipMalloc((void**)&d_wpart, (size_t)doubles*sizeof(double)));
    else MIMSEM_HIP_TRY(hipExtMallocWithFlags((void**)&d_wpart, (size_t)doubles*sizeof(double),
                                              w_partmem == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    wpart_doubles = doubles; bytes += doubles*8;
    return MIMSEM_OK;
}
int mimsem_ctx::ensure_wsplit(long long doubles) {
    if (doubles <= wsplit_doubles) return MIMSEM_OK;
    if (is_capturing() || split.pending) return MIMSEM_ERR_STATE;
    if (d_wsplit) { retired.push_back(d_wsplit); d_wsplit = nullptr; wsplit_doubles = 0; }
    MIMSEM_HIP_TRY(hipMalloc((void**)&d_wsplit, (size_t)doubles*sizeof(double)));
    wsplit_doubles = doubles; bytes += doubles*8;
    return MIMSEM_OK;
}
int mimsem_ctx::ensure_kry(long long doubles) {
    if (doubles <= kry_doubles) return MIMSEM_OK;
    if (is_capturing()) return MIMSEM_ERR_STATE;
    if (d_kry) { retired.push_back(d_kry); d_kry = nullptr; kry_doubles = 0; }
    MIMSEM_HIP_TRY(hipMalloc((void**)&d_kry, (size_t)doubles*sizeof(double)));
    kry_doubles = doubles; bytes += doubles*8;
    if (!d_rdcnt) {                                                  // arrival counters of the one-launch rowdot: zeroed once, every call leaves them zero
        MIMSEM_HIP_TRY(hipMalloc((void**)&d_rdcnt, MIMSEM_RD_COUNTERS*sizeof(unsigned)));
        MIMSEM_HIP_TRY(hipMemsetAsync(d_rdcnt, 0, MIMSEM_RD_COUNTERS*sizeof(unsigned), stream));
    }
    return MIMSEM_OK;
}
int mimsem_ctx::ensure_col(long long doubles) {
    if (doubles <= col_doubles) return MIMSEM_OK;
    // retired like d_ye / d_kry: a graph captured around a column call (Engine.capture accepts any fn) keeps the old address
    if (is_capturing()) return MIMSEM_ERR_STATE;          // growing = hipMalloc, illegal on a capturing stream: warm up outside the capture
    if (d_col) { retired.push_back(d_col); d_col = nullptr; col_doubles = 0; }
    MIMSEM_HIP_TRY(hipMalloc((void**)&d_col, (size_t)doubles*sizeof(double)));
    col_doubles = doubles; bytes += doubles*8;
    return MIMSEM_OK;
}

hipEvent_t mimsem_ctx::next_event() {
    if (ev_used == ev_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev_pool.push_back(e); }
    return ev_pool[ev_used++];
}

// (re)build the wave-level plan of a context from its host copies of the mesh; `marked` [n1] flags the 1-form slots that take part
// in a halo exchange (null: none).  Old device tables are retired, not freed (a captured graph may still hold them).
static int setup_wave(mimsem_ctx* c, const char* marked) {
    const ElemSizes& es = c->es;
    WavePlan P;
    const int lpe = es.mp12 <= 4 ? 4 : (es.mp12 <= 16 ? 16 : (es.mp12 <= 32 ? 32 : 64));
    const bool singles = exp_env("MIMSEM_WAVE_SINGLES") && atoi(exp_env("MIMSEM_WAVE_SINGLES")) != 0;
    // MIMSEM_WAVE_FIN=1: the experimental in-kernel finishing phase (side layout of the partial sums, no mixed pairs); default: round 2's
    // dense layout with mixed pairs + the perimeter pass (measured faster: DESIGN 4.6)
    const bool want_fin = exp_env("MIMSEM_WAVE_FIN") && atoi(exp_env("MIMSEM_WAVE_FIN")) != 0;
    const bool mixed = !want_fin && !(exp_env("MIMSEM_WAVE_MIXED") && atoi(exp_env("MIMSEM_WAVE_MIXED")) == 0);
    // round 5: TILE mode -- four wave-groups per workgroup, the partial sums of the slots they share meet in LDS behind ONE barrier per work
    // item instead of travelling through the workspace to the perimeter pass (MIMSEM_WAVE_TILE=0 | 1; orders 3 and 4, no halo split)
    const bool want_tile = (es.n == 3 || es.n == 4) && !marked && exp_env("MIMSEM_WAVE_TILE") && atoi(exp_env("MIMSEM_WAVE_TILE")) != 0;
    int rc = build_wave_plan(es.n, c->n1, c->nEl, es.n1e, es.n0e, 64/lpe, c->h_i1x.data(), c->h_i1y.data(), c->h_i0.data(), singles, mixed, marked, P, want_tile);
    if (rc && want_tile) {                                             // a numbering the tiles do not fit: the plain plan
        P = WavePlan();
        rc = build_wave_plan(es.n, c->n1, c->nEl, es.n1e, es.n0e, 64/lpe, c->h_i1x.data(), c->h_i1y.data(), c->h_i0.data(), singles, mixed, marked, P, false);
    }
    if (rc) return rc;
    void* old[] = {c->d_wlane, c->d_wplan, c->d_wprec, c->d_wnode, c->d_wsing, c->d_wG, c->d_wR, c->d_wfin, c->d_wsslot, c->d_wcnt, c->d_wtfin};
    c->d_wtfin = nullptr; c->w_ntiles = 0;
    for (void* p : old) if (p) c->retired.push_back(p);
    c->d_wlane = nullptr; c->d_wplan = nullptr; c->d_wprec = nullptr; c->d_wnode = nullptr; c->d_wsing = nullptr; c->d_wG = nullptr; c->d_wR = nullptr;
    c->d_wfin = nullptr; c->d_wsslot = nullptr; c->d_wcnt = nullptr;
    c->wave1 = false; c->w_fin = false;
    if ((rc = upload(&c->d_wlane, P.lane.data(), P.lane.size(), c))) return rc;
    if ((rc = upload(&c->d_wplan, P.plan.data(), P.plan.size(), c))) return rc;
    {
        std::vector<int4> rec(P.pslot.size());
        for (size_t i = 0; i < rec.size(); i++) rec[i] = int4{P.pslot[i], P.ppart[2*i], P.ppart[2*i + 1], 0};
        if ((rc = upload(&c->d_wprec, rec.data(), rec.size(), c))) return rc;
    }
    if ((rc = upload(&c->d_wnode, P.node.data(), P.node.size(), c))) return rc;
    if (P.nsing && (rc = upload(&c->d_wsing, P.sing.data(), P.sing.size(), c))) return rc;
    if ((rc = upload(&c->d_wfin, P.fin.data(), P.fin.size(), c))) return rc;
    if ((rc = upload(&c->d_wsslot, P.sslot.data(), P.sslot.size(), c))) return rc;
    if (P.ntiles > 0) { if ((rc = upload(&c->d_wtfin, P.tfin.data(), P.tfin.size(), c))) return rc; c->w_ntiles = P.ntiles; c->w_ninner = P.ninner; }
    {   // arrival counters of the finishing phase: one per (side, work item of a group), zero between launches
        const size_t n = (size_t)std::max(P.nsides, 1)*(size_t)std::max(c->nk, 1);
        MIMSEM_HIP_TRY(hipMalloc((void**)&c->d_wcnt, n*sizeof(int)));
        MIMSEM_HIP_TRY(hipMemset(c->d_wcnt, 0, n*sizeof(int)));
        c->bytes += (long long)(n*sizeof(int));
    }
    // packed metric of the wave kernel: {gaa, gab, gbb, 1/det} = Q/det J^T J per quadrature point (16-byte loads) and the
    // rotational factor (-J00 J11 + J01 J10) Q/det of RotMat, in wave-group order
    {
        std::vector<double> G((size_t)P.ngroups*64*4, 0.0), Rv((size_t)P.ngroups*64, 0.0);
        const int gsz = 64/lpe;
        for (int g = 0; g < P.ngroups; g++)
            for (int l = 0; l < 64; l++) {
                const int e = P.perm[(size_t)g*gsz + l/lpe], q = l%lpe;
                if (e < 0 || q >= es.mp12) continue;                 // padding element / lane beyond the point grid: zeros
                const double* Jq = c->h_J.data() + ((size_t)e*es.mp12 + q)*4;
                const double det = c->h_det[(size_t)e*es.mp12 + q];
                const double Q = c->tab.quad.w[q%es.mp1]*c->tab.quad.w[q/es.mp1];
                double* o = &G[((size_t)g*64 + l)*4];
                o[0] = (Jq[0]*Jq[0] + Jq[2]*Jq[2])*Q/det; o[1] = (Jq[0]*Jq[1] + Jq[2]*Jq[3])*Q/det;
                o[2] = (Jq[1]*Jq[1] + Jq[3]*Jq[3])*Q/det; o[3] = 1.0/det;
                Rv[(size_t)g*64 + l] = (-Jq[0]*Jq[3] + Jq[1]*Jq[2])*Q/det;
            }
        if ((rc = upload(&c->d_wG, G.data(), G.size(), c))) return rc;
        if ((rc = upload(&c->d_wR, Rv.data(), Rv.size(), c))) return rc;
    }
    c->w_ndirect = P.ndirect; c->w_ngroups = P.ngroups; c->w_nsing = P.nsing; c->w_nps = P.nps; c->w_npart = P.npart;
    c->w_npwritten = P.npwritten; c->w_nsides = P.nsides;
    if (const char* kind = exp_env("MIMSEM_WPART_MEM")) c->w_partmem = !strcmp(kind, "plain") ? 2 : (!strcmp(kind, "finegrained") ? 1 : 0);
    c->w_fin = P.fin_ok && want_fin;
    c->w_nbgroups = P.nbgroups; c->w_nbrec = P.nbrec; c->w_split = marked != nullptr; c->wave1 = true;
    if (const char* ev = exp_env("MIMSEM_WAVE_ORDER")) c->wave_order = atoi(ev);
    if (const char* ev = exp_env("MIMSEM_WAVE_LCH")) c->wave_lch = atoi(ev);
    if (const char* ev = exp_env("MIMSEM_WAVE_CPP")) c->wave_cpp = atoi(ev);
    if (const char* ev = exp_env("MIMSEM_WAVE2")) c->wave2_mode = atoi(ev);
    if (getenv("MIMSEM_VERBOSE"))
        fprintf(stderr, "[mimsem] wave plan: %d groups of %d elements (%d on the halo boundary), %d perimeter slots (%d partials in %d sides) of %d; "
                        "in-kernel finishing %s; tiles %d (%d inner slots finished in LDS, widest LDS row %d doubles)\n", P.ngroups, 64/lpe, P.nbgroups, P.nps,
                P.npwritten, P.nsides, c->n1, c->w_fin ? "on" : "off", P.ntiles, P.ninner, P.tpmax);
    return MIMSEM_OK;
}

extern "C" {

int mimsem_ctx_set_profiling(mimsem_ctx* c, int on) {
    if (!c) return MIMSEM_ERR_ARG;
    // on = n > 0: time every n-th mimsem_op_apply (n = 1: all of them); 0 = off
    c->profiling = on != 0; c->prof_every = on > 0 ? on : 1; c->prof_count = 0; c->ev_used = 0; c->ev_has2.clear();
    // the events of the first ~128 sampled applies are created HERE, not inside the caller's timed region (bench.py's headline region used to pay
    // a hipEventCreate x 4 per sampled step: its `value` sat 10-18 % under the median of the unprofiled repeats of the same region)
    if (on) { MIMSEM_HIP_TRY(hipSetDevice(c->device)); while (c->ev_pool.size() < 512) { hipEvent_t e; MIMSEM_HIP_TRY(hipEventCreate(&e)); c->ev_pool.push_back(e); } }
    return MIMSEM_OK;
}
int mimsem_ctx_profile_read(mimsem_ctx* c, double* ms1, double* ms2, long long* launches) {
    if (!c) return MIMSEM_ERR_ARG;
    MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream));
    double a = 0.0, b = 0.0;
    for (size_t i = 0; i + 3 < c->ev_used; i += 4) {
        float t = 0.f;
        MIMSEM_HIP_TRY(hipEventElapsedTime(&t, c->ev_pool[i], c->ev_pool[i + 1])); a += t;
        if (i/4 < c->ev_has2.size() && c->ev_has2[i/4]) {                                          // only when THIS apply recorded its second pair
            if (hipEventElapsedTime(&t, c->ev_pool[i + 2], c->ev_pool[i + 3]) == hipSuccess) b += t;
            else (void)hipGetLastError();
        }
    }
    if (ms1) *ms1 = a;
    if (ms2) *ms2 = b;
    if (launches) *launches = (long long)(c->ev_used/4);
    c->ev_used = 0; c->ev_has2.clear();
    return MIMSEM_OK;
}

int mimsem_abi_version(void) { return MIMSEM_ABI_VERSION; }
int mimsem_build_has_experiments(void) { return kExperiments ? 1 : 0; }

const char* mimsem_strerror(int code) {
    switch (code) {
    case MIMSEM_OK: return "ok";
    case MIMSEM_ERR_ARG: return "invalid argument";
    case MIMSEM_ERR_UNSUPPORTED: return "unsupported configuration (order must be 1..7 with quadrature order == element order)";
    case MIMSEM_ERR_HIP: return "HIP runtime error";
    case MIMSEM_ERR_STATE: return "context state does not allow this call";
    case MIMSEM_ERR_SINGULAR: return "singular block";
    }
    return "unknown error";
}
const char* mimsem_last_hip_error(void) { return mimsem::g_last_hip_error.c_str(); }

int mimsem_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mimsem_ctx_create(const mimsem_mesh_desc* d, int device, mimsem_ctx** out) {
    if (!d || !out) return MIMSEM_ERR_ARG;
    *out = nullptr;
    if (d->elOrd < 1 || d->elOrd > 7 || d->quadOrd != d->elOrd) return MIMSEM_ERR_UNSUPPORTED;
    if (d->nEl < 0 || d->nk < 1 || d->n0 < 0 || d->n1 < 0 || d->n2 < 0) return MIMSEM_ERR_ARG;
    if (!d->inds0 || !d->inds1x || !d->inds1y || !d->det || !d->J) return MIMSEM_ERR_ARG;
    MIMSEM_HIP_TRY(hipSetDevice(device));

    mimsem_ctx* c = new mimsem_ctx();
    c->device = device;
    if (const char* ev = exp_env("MIMSEM_LCH")) c->lch_override = atoi(ev);
    if (const char* ev = exp_env("MIMSEM_NOSWZ")) c->swz = atoi(ev) ? 0 : 1;
    c->es = ElemSizes(d->elOrd);
    c->nEl = d->nEl; c->nk = d->nk; c->n0 = d->n0; c->n1 = d->n1; c->n2 = d->n2;
    const ElemSizes& es = c->es;
    if (!c->tab.init(d->elOrd, d->quadOrd) || !c->tab.collocated) { delete c; return MIMSEM_ERR_UNSUPPORTED; }
    if (d->n2 < d->nEl*es.n2e && !d->inds2) { delete c; return MIMSEM_ERR_ARG; }

    int rc = MIMSEM_OK;
    auto fail = [&](int code) { mimsem_ctx_destroy(c); return code; };
    if ((rc = upload(&c->d_E, c->tab.E.data(), c->tab.E.size(), c))) return fail(rc);
    if ((rc = upload(&c->d_w, c->tab.quad.w.data(), c->tab.quad.w.size(), c))) return fail(rc);
    if ((rc = upload(&c->d_xn, c->tab.nodes.x.data(), c->tab.nodes.x.size(), c))) return fail(rc);
    if ((rc = upload(&c->d_U, c->tab.U.data(), c->tab.U.size(), c))) return fail(rc);
    if ((rc = upload(&c->d_V, c->tab.V.data(), c->tab.V.size(), c))) return fail(rc);
    if ((rc = upload(&c->d_W, c->tab.W.data(), c->tab.W.size(), c))) return fail(rc);
    if ((rc = upload(&c->d_P, c->tab.P.data(), c->tab.P.size(), c))) return fail(rc);

    // metric: [e][q][4] -> component-major [e][4][q] so that one lane-per-point load is contiguous
    {
        std::vector<double> Js((size_t)d->nEl*4*es.mp12);
        for (int e = 0; e < d->nEl; e++)
            for (int q = 0; q < es.mp12; q++)
                for (int k = 0; k < 4; k++)
                    Js[((size_t)e*4 + k)*es.mp12 + q] = d->J[((size_t)e*es.mp12 + q)*4 + k];
        if ((rc = upload(&c->d_J, Js.data(), Js.size(), c))) return fail(rc);
    }
    if ((rc = upload(&c->d_det, d->det, (size_t)d->nEl*es.mp12, c))) return fail(rc);
    if ((rc = upload(&c->d_i0, d->inds0, (size_t)d->nEl*es.n0e, c))) return fail(rc);
    if ((rc = upload(&c->d_i1x, d->inds1x, (size_t)d->nEl*es.n1e, c))) return fail(rc);
    if ((rc = upload(&c->d_i1y, d->inds1y, (size_t)d->nEl*es.n1e, c))) return fail(rc);
    if (d->inds2) {
        bool contig = true;
        for (size_t i = 0; i < (size_t)d->nEl*es.n2e && contig; i++) contig = (d->inds2[i] == (int)i);
        for (size_t i = 0; i < (size_t)d->nEl*es.n2e; i++) if (d->inds2[i] < 0 || d->inds2[i] >= d->n2) return fail(MIMSEM_ERR_ARG);
        c->inds2_contig = contig;
        if (!contig && (rc = upload(&c->d_i2, d->inds2, (size_t)d->nEl*es.n2e, c))) return fail(rc);
    }

    if (d->indsq) {
        if (d->nq <= 0) return fail(MIMSEM_ERR_ARG);
        for (size_t i = 0; i < (size_t)d->nEl*es.mp12; i++) if (d->indsq[i] < 0 || d->indsq[i] >= d->nq) return fail(MIMSEM_ERR_ARG);
        c->nq = d->nq;
        if ((rc = upload(&c->d_iq, d->indsq, (size_t)d->nEl*es.mp12, c))) return fail(rc);
    }
    // scatter-add plans
    {
        std::vector<int> plan;
        rc = build_plan(d->n1, d->nEl, 2*es.n1e, {d->inds1x, d->inds1y}, {0, es.n1e}, es.n1e, 2, plan);
        if (rc) return fail(rc);
        if ((rc = upload(&c->d_g1, plan.data(), plan.size(), c))) return fail(rc);
        {
            // the block passes' own view of that plan (round 6, late): per (element, block row) {slot, first contributor, second contributor}
            // as ONE 16-byte entry -- the row's chain of dependent loads is {this entry -> the gathered values} instead of {edge map -> plan ->
            // values}; at the ~3 500 elements of the shallow-water drivers a block pass IS its chain of memory latencies
            const int nd = 2*es.n1e;
            std::vector<int> bp((size_t)d->nEl*nd*4, 0);
            for (int e = 0; e < d->nEl; e++)
                for (int r = 0; r < nd; r++) {
                    const int slot = r < es.n1e ? d->inds1x[(size_t)e*es.n1e + r] : d->inds1y[(size_t)e*es.n1e + r - es.n1e];
                    int* o = &bp[((size_t)e*nd + r)*4];
                    o[0] = slot; o[1] = plan[(size_t)slot*2]; o[2] = plan[(size_t)slot*2 + 1]; o[3] = 0;
                }
            if ((rc = upload(&c->d_bplan, bp.data(), bp.size(), c))) return fail(rc);
        }
        c->G0 = 4;
        rc = build_plan(d->n0, d->nEl, es.n0e, {d->inds0}, {0}, es.n0e, 4, plan);
        if (rc == MIMSEM_ERR_UNSUPPORTED) { c->G0 = 8; rc = build_plan(d->n0, d->nEl, es.n0e, {d->inds0}, {0}, es.n0e, 8, plan); }
        if (rc) return fail(rc);
        if ((rc = upload(&c->d_g0, plan.data(), plan.size(), c))) return fail(rc);
    }
    // direct-write tables: a DoF with a single contributing element needs no scatter-add at all
    {
        auto build = [&](int nslots, const std::vector<const int*>& maps, int counts, std::vector<std::vector<int>>& dir, std::vector<int>& shared) {
            std::vector<int> cnt(nslots, 0);
            for (const int* m : maps) for (size_t i = 0; i < (size_t)d->nEl*counts; i++) cnt[m[i]]++;
            dir.assign(maps.size(), std::vector<int>((size_t)d->nEl*counts));
            for (size_t k = 0; k < maps.size(); k++)
                for (size_t i = 0; i < (size_t)d->nEl*counts; i++) dir[k][i] = (cnt[maps[k][i]] == 1) ? maps[k][i] : -1;
            shared.clear();
            for (int sl = 0; sl < nslots; sl++) if (cnt[sl] != 1) shared.push_back(sl);     // incl. untouched slots (written as 0)
        };
        std::vector<std::vector<int>> dir; std::vector<int> sh;
        build(d->n1, {d->inds1x, d->inds1y}, es.n1e, dir, sh);
        if ((rc = upload(&c->d_d1x, dir[0].data(), dir[0].size(), c))) return fail(rc);
        if ((rc = upload(&c->d_d1y, dir[1].data(), dir[1].size(), c))) return fail(rc);
        if ((rc = upload(&c->d_sh1, sh.data(), sh.size(), c))) return fail(rc);
        c->nsh1 = (int)sh.size();
        build(d->n0, {d->inds0}, es.n0e, dir, sh);
        if ((rc = upload(&c->d_d0, dir[0].data(), dir[0].size(), c))) return fail(rc);
        if ((rc = upload(&c->d_sh0, sh.data(), sh.size(), c))) return fail(rc);
        c->nsh0 = (int)sh.size();
        // measured SLOWER on MI355X (pass 1 14.9 -> 17.4 us: the single-contributor stores are scattered, the ye stores coalesce;
        // pass 2 over the shared-slot list 7.4 -> 8.2 us), profiles/r01_direct_interior_ab.txt: opt-in only
        c->direct = exp_env("MIMSEM_DIRECT") != nullptr;
    }
    // Fused single-barrier scatter-add (group-local sums in LDS + perimeter pass): measured SLOWER than the two-pass
    // form on MI355X in round 1 (profiles/r01_fused_scatter_ab.txt), so it is opt-in (MIMSEM_FUSE=1) for further tuning.
    if (exp_env("MIMSEM_FUSE")) {
        FusedPlan P;
        const int epb = 256/(es.mp12 <= 4 ? 4 : (es.mp12 <= 16 ? 16 : (es.mp12 <= 32 ? 32 : 64)));
        rc = build_fused_plan(d->n1, d->nEl, es.n1e, epb, d->inds1x, d->inds1y, P);
        if (rc == MIMSEM_OK && P.lmax < 0x8000) {
            if ((rc = upload(&c->d_fperm, P.perm.data(), P.perm.size(), c))) return fail(rc);
            if ((rc = upload(&c->d_flid, P.lid.data(), P.lid.size(), c))) return fail(rc);
            if ((rc = upload(&c->d_fslot, P.fslot.data(), P.fslot.size(), c))) return fail(rc);
            if ((rc = upload(&c->d_fcnt, P.fcnt.data(), P.fcnt.size(), c))) return fail(rc);
            if ((rc = upload(&c->d_pslot, P.pslot.data(), P.pslot.size(), c))) return fail(rc);
            if ((rc = upload(&c->d_ppart, P.ppart.data(), P.ppart.size(), c))) return fail(rc);
            c->f_ngroups = P.ngroups; c->f_lmax = P.lmax; c->f_nps = P.nps; c->f_npart = P.npart; c->fused1 = true;
            if (getenv("MIMSEM_VERBOSE")) {
                long long tot = 0; for (int v : P.fcnt) tot += v;
                fprintf(stderr, "[mimsem] fused plan: %d groups of %d elements, %.1f local slots/group, %d perimeter slots (%d partials) of %d\n",
                        P.ngroups, epb, (double)tot/std::max(P.ngroups, 1), P.nps, P.npart, d->n1);
            }
        }
    }
    // Wave-level fused scatter-add of the 1-form -> 1-form operators (k_apply_wave): the default since round 2
    // (MIMSEM_WAVE=0 selects the two-pass form for every operator).  Host copies of the index maps and the metric stay with the
    // context: mimsem_ctx_set_halo_slots re-derives the plan with the boundary groups first.
    // block pass of the Chebyshev / Richardson sweeps on the matrix cores: the default at p = 4 (40 x 40 blocks: 32.1 us against 32.8 us for
    // the register-row form on the config-5 grid, profiles/r03_mfma_p4_ab.txt), off at p <= 3 (24 x 24: 25.3 against 23.8 us);
    // MIMSEM_BLOCKS_MFMA=0 | 1 overrides
    c->cheb_pend = exp_env("MIMSEM_CHEB_PEND") && atoi(exp_env("MIMSEM_CHEB_PEND")) != 0;
    c->blocks_mfma = exp_env("MIMSEM_BLOCKS_MFMA") ? atoi(exp_env("MIMSEM_BLOCKS_MFMA")) != 0 : es.n == 4;
    if (d->nEl > 0) { c->h_e1x.assign(d->inds1x, d->inds1x + (size_t)d->nEl*es.n1e); c->h_e1y.assign(d->inds1y, d->inds1y + (size_t)d->nEl*es.n1e);
                      c->h_e0.assign(d->inds0, d->inds0 + (size_t)d->nEl*es.n0e); }
    c->memset_node = exp_env("MIMSEM_MEMSET_NODE") && atoi(exp_env("MIMSEM_MEMSET_NODE")) != 0;      // (read once: mimsem_memset is a hot call of the recorded solves)
    c->rd_two = exp_env("MIMSEM_ROWDOT_TWO") && atoi(exp_env("MIMSEM_ROWDOT_TWO")) != 0;
    c->blu_stop = exp_env("MIMSEM_BLU_STOP") ? atoi(exp_env("MIMSEM_BLU_STOP")) : 0;
    c->pivot_fallback = getenv("MIMSEM_COLUMN_PIVOT_FALLBACK") ? std::min(2, std::max(0, atoi(getenv("MIMSEM_COLUMN_PIVOT_FALLBACK")))) : 1;      // on by default since round 5 (mimsem_column_set_pivot_fallback)
    if (!(getenv("MIMSEM_WAVE") && atoi(getenv("MIMSEM_WAVE")) == 0) && !c->fused1 && !c->direct && d->nEl > 0 && es.n <= 4) {
        c->h_i1x.assign(d->inds1x, d->inds1x + (size_t)d->nEl*es.n1e); c->h_i1y.assign(d->inds1y, d->inds1y + (size_t)d->nEl*es.n1e);
        c->h_i0.assign(d->inds0, d->inds0 + (size_t)d->nEl*es.n0e);
        c->h_J.assign(d->J, d->J + (size_t)d->nEl*es.mp12*4); c->h_det.assign(d->det, d->det + (size_t)d->nEl*es.mp12);
        rc = setup_wave(c, nullptr);
        if (rc == MIMSEM_ERR_ARG || rc == MIMSEM_ERR_HIP) return fail(rc);
        rc = MIMSEM_OK;
    }
    {
        const size_t cnt = (size_t)d->nk*d->nEl*es.mp12;
        hipError_t he = hipMalloc((void**)&c->d_th, std::max<size_t>(cnt, 1)*sizeof(double));
        if (he == hipSuccess) he = hipMalloc((void**)&c->d_tI, std::max<size_t>(cnt, 1)*sizeof(double));
        const size_t cntp = 2*(size_t)(d->nk/2 + 1)*d->nEl*es.mp12*2;
        if (he == hipSuccess && es.n <= 4) he = hipMalloc((void**)&c->d_tIp, std::max<size_t>(cntp, 1)*sizeof(double));
        if (he != hipSuccess) return fail(mimsem::hip_fail(he, "hipMalloc(thickness)"));
        c->bytes += 2*(long long)cnt*8 + (es.n <= 4 ? (long long)cntp*8 : 0);
    }
    if ((rc = mimsem_ctx_set_levels(c, d->thick, d->thickInv))) return fail(rc);
    // two element-local buffers + one packed [1-form | 2-form] row per level: the largest request of any entry point at nlev <= nk
    if ((rc = c->ensure_ye((long long)d->nk*(2LL*d->nEl*std::max(2*es.n1e, es.n0e) + d->n1 + d->n2)))) return fail(rc);
    *out = c;
    return MIMSEM_OK;
}

struct mimsem_graph;
static void orphan_graphs(mimsem_ctx* c);
void mimsem_ctx_destroy(mimsem_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    orphan_graphs(c);
    void* ptrs[] = {c->d_xn, c->d_E, c->d_w, c->d_U, c->d_V, c->d_W, c->d_P, c->d_J, c->d_det, c->d_th, c->d_tI, c->d_tIp, c->d_tIn,
                    c->d_i0, c->d_i1x, c->d_i1y, c->d_i2, c->d_iq, c->d_fperm, c->d_flid, c->d_fslot, c->d_fcnt, c->d_pslot, c->d_ppart, c->d_wlane, c->d_wplan, c->d_wprec, c->d_wnode, c->d_wsing, c->d_wG, c->d_wR, c->d_wfin, c->d_wsslot, c->d_wcnt, c->d_wtfin, c->d_wpart, c->d_wsplit, c->d_colstat, c->d_forceflag, c->d_rdcnt, c->d_cheb, c->d_colratio, c->d_g1, c->d_bplan, c->d_g0, c->d_ye, c->d_col, c->d_lu, c->d_kry,
                    c->d_d0, c->d_d1x, c->d_d1y, c->d_sh0, c->d_sh1};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (void* p : c->retired) (void)hipFree(p);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    delete c;
}

int mimsem_ctx_set_stream(mimsem_ctx* c, void* s) {
    if (!c) return MIMSEM_ERR_ARG;
    if (c->cap_active) return MIMSEM_ERR_STATE;                          // (between mimsem_graph_begin and _end the stream is the capture's)
    c->stream = (hipStream_t)s;
    return MIMSEM_OK;
}
// a stream of the context's own for hosts that cannot create one (no HIP headers): non-blocking, so nothing this context launches orders itself
// against the legacy default stream any more
int mimsem_ctx_use_own_stream(mimsem_ctx* c) {
    if (!c) return MIMSEM_ERR_ARG;
    if (c->cap_active) return MIMSEM_ERR_STATE;
    MIMSEM_HIP_TRY(hipSetDevice(c->device));
    if (!c->own_stream) MIMSEM_HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    return MIMSEM_OK;
}
// ---- hipGraph capture for hosts without a HIP toolchain (include/mimsem_hip.h) -----------------------------------------------------
// A recording belongs to the context it was made on (advisor, round 5): it replays on the context's CURRENT stream -- so that "in stream order
// with the context's other work" stays true after mimsem_ctx_set_stream / _use_own_stream -- except that a recording made on the capture
// stream a default-stream context was lent keeps that stream (the legacy default stream cannot launch graphs in order with it either way);
// mimsem_ctx_destroy orphans the recordings still alive: their launch returns MIMSEM_ERR_STATE instead of using a destroyed stream.
struct mimsem_graph { hipGraph_t g = nullptr; hipGraphExec_t x = nullptr; hipStream_t stream = nullptr; int nodes = 0; mimsem_ctx* c = nullptr; bool lent = false; };
int mimsem_graph_begin(mimsem_ctx* c) {
    if (!c) return MIMSEM_ERR_ARG;
    if (c->cap_active || c->is_capturing()) return MIMSEM_ERR_STATE;
    MIMSEM_HIP_TRY(hipSetDevice(c->device));
    c->cap_swapped = false;
    if (c->stream == nullptr) {                                          // the legacy default stream cannot capture: a blocking stream of the
        if (!c->cap_stream) MIMSEM_HIP_TRY(hipStreamCreateWithFlags(&c->cap_stream, hipStreamDefault));   // context's own keeps its order with it
        c->cap_saved = c->stream; c->stream = c->cap_stream; c->cap_swapped = true;
    }
    const hipError_t he = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
    if (he != hipSuccess) {
        if (c->cap_swapped) { c->stream = c->cap_saved; c->cap_swapped = false; }
        return mimsem::hip_fail(he, "hipStreamBeginCapture");
    }
    c->cap_active = true;
    return MIMSEM_OK;
}
int mimsem_graph_end(mimsem_ctx* c, mimsem_graph** out) {
    if (!c || !out) return MIMSEM_ERR_ARG;
    if (!c->cap_active) return MIMSEM_ERR_STATE;
    *out = nullptr;
    hipGraph_t g = nullptr;
    const hipError_t he = hipStreamEndCapture(c->stream, &g);
    const hipStream_t used = c->stream;
    c->cap_active = false;
    if (c->cap_swapped) { c->stream = c->cap_saved; c->cap_swapped = false; }
    if (he != hipSuccess || !g) { (void)hipGetLastError(); return he == hipSuccess ? MIMSEM_ERR_STATE : mimsem::hip_fail(he, "hipStreamEndCapture"); }
    mimsem_graph* gr = new mimsem_graph();
    gr->g = g; gr->stream = used; gr->c = c; gr->lent = used != c->stream;
    size_t nn = 0;
    if (hipGraphGetNodes(g, nullptr, &nn) == hipSuccess) gr->nodes = (int)nn;
    const hipError_t hi = hipGraphInstantiate(&gr->x, g, nullptr, nullptr, 0);
    if (hi != hipSuccess) { (void)hipGraphDestroy(g); delete gr; return mimsem::hip_fail(hi, "hipGraphInstantiate"); }
    c->graphs.push_back(gr);
    *out = gr;
    return MIMSEM_OK;
}
int mimsem_graph_launch(mimsem_graph* g) {
    if (!g || !g->x) return MIMSEM_ERR_ARG;
    if (!g->c) return MIMSEM_ERR_STATE;                                  // its context is gone
    MIMSEM_HIP_TRY(hipGraphLaunch(g->x, (g->lent || g->c->stream == nullptr) ? g->stream : g->c->stream));
    return MIMSEM_OK;
}
int mimsem_graph_num_nodes(const mimsem_graph* g) { return g ? g->nodes : MIMSEM_ERR_ARG; }
static void orphan_graphs(mimsem_ctx* c) { for (void* p : c->graphs) ((mimsem_graph*)p)->c = nullptr; c->graphs.clear(); }
void mimsem_graph_destroy(mimsem_graph* g) {
    if (!g) return;
    if (g->c) { auto& v = g->c->graphs; v.erase(std::remove(v.begin(), v.end(), (void*)g), v.end()); }
    if (g->x) (void)hipGraphExecDestroy(g->x);
    if (g->g) (void)hipGraphDestroy(g->g);
    delete g;
}
int mimsem_ctx_sync(mimsem_ctx* c) { if (!c) return MIMSEM_ERR_ARG; MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream)); return MIMSEM_OK; }
long long mimsem_ctx_workspace_bytes(const mimsem_ctx* c) { return c ? c->bytes : 0; }
int mimsem_op_level_chunk(const mimsem_ctx* c, int nlev) { return (c && nlev > 0) ? level_chunk(c, nlev) : MIMSEM_ERR_ARG; }
int mimsem_op_wave_stats(const mimsem_ctx* c, int nlev, int out[5]) {
    if (!c || !out || nlev < 1) return MIMSEM_ERR_ARG;
    for (int i = 0; i < 5; i++) out[i] = 0;
    if (!c->wave1) return 0;
    out[0] = c->w_ngroups; out[1] = c->w_ndirect; out[2] = c->w_npwritten; out[3] = c->w_nps;
    { const int lch = wave_level_chunk(c, nlev); out[4] = lch*wave_chunks_per_item(c, nlev, lch, c->w_ngroups); }
    return 1;
}

int mimsem_ctx_set_levels(mimsem_ctx* c, const double* thick, const double* thickInv) {
    if (!c) return MIMSEM_ERR_ARG;
    if (c->is_capturing()) return MIMSEM_ERR_STATE;
    MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream));        // (a non-blocking stream does not order itself against the synchronous copies below)
    const size_t cnt = (size_t)c->nk*c->nEl*c->es.mp12;
    std::vector<double> th(cnt, 1.0), ti(cnt, 1.0);
    if (thick) std::memcpy(th.data(), thick, cnt*sizeof(double));
    if (thickInv) std::memcpy(ti.data(), thickInv, cnt*sizeof(double));
    else if (thick) for (size_t i = 0; i < cnt; i++) ti[i] = 1.0/th[i];     // Geom.cpp:761
    else if (false) {}
    if (cnt) {
        MIMSEM_HIP_TRY(hipMemcpy(c->d_th, th.data(), cnt*sizeof(double), hipMemcpyHostToDevice));
        MIMSEM_HIP_TRY(hipMemcpy(c->d_tI, ti.data(), cnt*sizeof(double), hipMemcpyHostToDevice));
    }
    if (cnt && c->d_tIp) {
        // level pairs for the wave-level kernels: entry (parity, m) holds levels L = 2m + parity and L + 1 (beyond the last level: the last again)
        const size_t row = (size_t)c->nEl*c->es.mp12, np = (size_t)c->nk/2 + 1;
        std::vector<double> tp(2*np*row*2);
        for (int par = 0; par < 2; par++) for (size_t m = 0; m < np; m++) {
            const size_t L0 = std::min<size_t>(2*m + par, c->nk - 1), L1 = std::min<size_t>(2*m + par + 1, c->nk - 1);
            double* o = tp.data() + ((size_t)par*np + m)*row*2;
            const double *a0 = ti.data() + L0*row, *a1 = ti.data() + L1*row;
            for (size_t i = 0; i < row; i++) { o[2*i] = a0[i]; o[2*i + 1] = a1[i]; }
        }
        MIMSEM_HIP_TRY(hipMemcpy(c->d_tIp, tp.data(), tp.size()*sizeof(double), hipMemcpyHostToDevice));
        // the same table per NODE (experiment, MIMSEM_WAVE_TNODE=1): the reference's thickInv lives on the nodes (eul/Geom.cpp:143-146, :761) and
        // the ABI takes it gathered per element: 16 values per element at p = 3 of which 9 are distinct per element on average.  Only when
        // every element holds the same bits at a shared node (and quadrature points are the nodes: mp12 == n0e)
        if (c->d_tIn) { (void)hipFree(c->d_tIn); c->d_tIn = nullptr; }
        if (exp_env("MIMSEM_WAVE_TNODE") && atoi(exp_env("MIMSEM_WAVE_TNODE")) == 1 && c->es.mp12 == c->es.n0e && !c->h_i0.empty() && c->n0 > 0) {
            const size_t n0 = (size_t)c->n0;
            std::vector<double> nod((size_t)c->nk*n0, 1.0), tn(2*np*n0*2, 1.0);
            std::vector<unsigned char> seen(n0);
            bool same = true;
            for (size_t L = 0; L < (size_t)c->nk && same; L++) {
                std::fill(seen.begin(), seen.end(), 0);
                const double* a0 = ti.data() + L*row;
                double* nl = nod.data() + L*n0;
                for (size_t i = 0; i < row; i++) {
                    const size_t nd = (size_t)c->h_i0[i];
                    if (nd >= n0 || (seen[nd] && std::memcmp(&nl[nd], &a0[i], 8))) { same = false; break; }
                    nl[nd] = a0[i]; seen[nd] = 1;
                }
            }
            for (int par = 0; par < 2 && same; par++) for (size_t m = 0; m < np; m++) {
                const size_t L0 = std::min<size_t>(2*m + par, c->nk - 1), L1 = std::min<size_t>(2*m + par + 1, c->nk - 1);
                double* o = tn.data() + ((size_t)par*np + m)*n0*2;
                for (size_t nd = 0; nd < n0; nd++) { o[2*nd] = nod[L0*n0 + nd]; o[2*nd + 1] = nod[L1*n0 + nd]; }
            }
            if (same) {
                MIMSEM_HIP_TRY(hipMalloc((void**)&c->d_tIn, tn.size()*sizeof(double)));
                MIMSEM_HIP_TRY(hipMemcpy(c->d_tIn, tn.data(), tn.size()*sizeof(double), hipMemcpyHostToDevice));
            }
        }
    }
    c->have_levels = (thick != nullptr) || (thickInv != nullptr);
    return MIMSEM_OK;
}

int mimsem_malloc(void** dev, long long bytes) {
    if (!dev || bytes < 0) return MIMSEM_ERR_ARG;
    MIMSEM_HIP_TRY(hipMalloc(dev, (size_t)std::max<long long>(bytes, 8)));
    return MIMSEM_OK;
}
int mimsem_free(void* dev) { if (dev) MIMSEM_HIP_TRY(hipFree(dev)); return MIMSEM_OK; }
int mimsem_memcpy_h2d(mimsem_ctx* c, void* dev, const void* host, long long bytes) {
    if (!c || !dev || !host || bytes < 0) return MIMSEM_ERR_ARG;
    MIMSEM_HIP_TRY(hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream));
    return MIMSEM_OK;
}
int mimsem_memcpy_d2h(mimsem_ctx* c, void* host, const void* dev, long long bytes) {
    if (!c || !dev || !host || bytes < 0) return MIMSEM_ERR_ARG;
    if (bytes > 0 && bytes <= 4096 && !c->is_capturing()) {                   // a handful of scalars (check norms, counters): through pinned memory
        if (!c->h_pin) MIMSEM_HIP_TRY(hipHostMalloc(&c->h_pin, 4096, hipHostMallocDefault));
        MIMSEM_HIP_TRY(hipMemcpyAsync(c->h_pin, dev, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
        MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(host, c->h_pin, (size_t)bytes);
        return MIMSEM_OK;
    }
    MIMSEM_HIP_TRY(hipMemcpyAsync(host, dev, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream));
    return MIMSEM_OK;
}
namespace {
__global__ __launch_bounds__(256) void k_fill64(long long n, unsigned long long v, unsigned long long* __restrict__ out) {
    const long long i = (long long)blockIdx.x*256 + threadIdx.x;
    if (i < n) out[i] = v;
}
}  // namespace
// (8-byte-aligned fills go through a kernel: inside a recorded graph a memset NODE costs tens of microseconds on this runtime, a kernel node
// two or three -- the fixed-length solves of the shallow-water step clear five vectors per Picard iteration)
int mimsem_memset(mimsem_ctx* c, void* dev, int byte, long long bytes) {
    if (!c || !dev || bytes < 0) return MIMSEM_ERR_ARG;
    if (bytes == 0) return MIMSEM_OK;
    if (bytes % 8 == 0 && ((uintptr_t)dev & 7u) == 0 && !c->memset_node) {
        const unsigned long long v = 0x0101010101010101ull*(unsigned long long)(unsigned char)byte;
        const long long n = bytes/8;
        hipLaunchKernelGGL(k_fill64, dim3((unsigned)((n + 255)/256)), dim3(256), 0, c->stream, n, v, (unsigned long long*)dev);
        MIMSEM_HIP_TRY(hipGetLastError());
        return MIMSEM_OK;
    }
    MIMSEM_HIP_TRY(hipMemsetAsync(dev, byte, (size_t)bytes, c->stream));
    return MIMSEM_OK;
}

// ---- horizontal operators --------------------------------------------------------------------
static int op_apply_core(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                         const double* f, long long fs, const double* f2, long long f2s, double param,
                         const double* x, long long xs, double* y, long long ys, double alpha,
                         const GatherEpilogue* epi = nullptr, const double* blocks = nullptr, int part = 0);
static bool is_up_op(int op);

int mimsem_op_apply(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                    const double* f, long long fs, const double* x, long long xs,
                    double* y, long long ys, double alpha) {
    if (is_up_op(op)) return MIMSEM_ERR_ARG;   // need mimsem_op_apply_up
    return op_apply_core(c, op, geom_lev0, nlev, scale, flags, f, fs, nullptr, 0, 0.0, x, xs, y, ys, alpha);
}

static bool is_up_op(int op) {
    return op == MIMSEM_OP_PHMAT_UP || op == MIMSEM_OP_ROTMAT_UP || op == MIMSEM_OP_UMAT_UP || op == MIMSEM_OP_UHMAT_UP ||
           op == MIMSEM_OP_UVEC_HU_UP || op == MIMSEM_OP_UMAT_RAY;
}
int mimsem_ctx_set_halo_slots(mimsem_ctx* c, int form, const int* slots, int n) {
    if (!c || form != 1 || n < 0 || (n && !slots)) return MIMSEM_ERR_ARG;
    if (c->is_capturing() || c->split.pending) return MIMSEM_ERR_STATE;      // (a pending BOUNDARY part belongs to the plan in force)
    if (!c->wave1 && c->h_i1x.empty()) {      // two-pass form: nothing to reorder (the split degenerates, see mimsem_op_apply_part); the marks are kept all the same
        c->h_halo1.assign(c->n1, 0);
        for (int i = 0; i < n; i++) { if (slots[i] < 0 || slots[i] >= c->n1) return MIMSEM_ERR_ARG; c->h_halo1[slots[i]] = 1; }
        return MIMSEM_OK;
    }
    std::vector<char> marked(std::max(c->n1, 1), 0);
    for (int i = 0; i < n; i++) { if (slots[i] < 0 || slots[i] >= c->n1) return MIMSEM_ERR_ARG; marked[slots[i]] = 1; }
    c->h_halo1.assign(marked.begin(), marked.begin() + c->n1);        // (kept: the block preconditioners weight shared edges by their GLOBAL multiplicity, csrc/ksp.hip)
    MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream));
    const int rc = setup_wave(c, marked.data());
    return rc == MIMSEM_ERR_UNSUPPORTED ? MIMSEM_OK : rc;      // numbering without a wave-level plan: the two-pass form stays, the split degenerates
}

int mimsem_op_apply_part(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                         const double* f, long long fs, const double* x, long long xs,
                         double* y, long long ys, double alpha, int part) {
    if (is_up_op(op) || part < MIMSEM_PART_ALL || part > MIMSEM_PART_INTERIOR) return MIMSEM_ERR_ARG;
    return op_apply_core(c, op, geom_lev0, nlev, scale, flags, f, fs, nullptr, 0, 0.0, x, xs, y, ys, alpha, nullptr, nullptr, part);
}

int mimsem_op_apply_part_reset(mimsem_ctx* c) {
    if (!c) return MIMSEM_ERR_ARG;
    c->split.pending = false;
    return MIMSEM_OK;
}

int mimsem_op_apply_up(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, double tau, unsigned flags,
                       const double* f, long long fs, const double* u, long long us,
                       const double* x, long long xs, double* y, long long ys, double alpha) {
    if (!is_up_op(op) || !u) return MIMSEM_ERR_ARG;
    if ((flags & MIMSEM_FLAG_TRANSPOSE) && op != MIMSEM_OP_UMAT_UP && op != MIMSEM_OP_UHMAT_UP) return MIMSEM_ERR_ARG;
    return op_apply_core(c, op, geom_lev0, nlev, scale, flags, f, fs, u, us, tau, x, xs, y, ys, alpha);
}

}  // extern "C"

static int op_apply_core(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                         const double* f, long long fs, const double* f2, long long f2s, double param,
                         const double* x, long long xs, double* y, long long ys, double alpha,
                         const GatherEpilogue* epi, const double* blocks, int part) {
    if (!c || nlev < 0) return MIMSEM_ERR_ARG;
    // interior / boundary split: only the wave-level form with marked halo slots really splits; everything else runs whole as
    // "the boundary part" and has nothing left for "the interior part", so callers can always issue both
    const bool wave_op = op == MIMSEM_OP_UMAT || op == MIMSEM_OP_UHMAT || op == MIMSEM_OP_ROTMAT || op == MIMSEM_OP_UTMAT || op == MIMSEM_OP_UTMAT_H;
    const bool splits = part != 0 && c->wave1 && c->w_split && wave_op && !epi &&
                        (long long)c->n1 < (1LL << 28) && (long long)c->n0 < (1LL << 28) && (long long)c->n2 < (1LL << 28) && (long long)c->nEl*c->es.mp12 < (1LL << 28);
    if (part == MIMSEM_PART_INTERIOR && !splits) return MIMSEM_OK;
    int in, cf, outsp;
    if (op_spaces(op, &in, &cf, &outsp)) return MIMSEM_ERR_ARG;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;       // empty batch: nothing to do (pointers of empty arrays may be null)
    if (!x || !y) return MIMSEM_ERR_ARG;
    if (cf >= 0 && !f) return MIMSEM_ERR_ARG;
    if (geom_lev0 < 0 || geom_lev0 + nlev > c->nk) return MIMSEM_ERR_ARG;
    if (op == MIMSEM_OP_UTMAT && geom_lev0 + nlev > c->nk - 1) return MIMSEM_ERR_ARG;   // needs thick[lev+1]
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (op == MIMSEM_OP_WMATINV || op == MIMSEM_OP_WHMATINV)
        return mimsem_colop_block_inverse_apply(c, op, geom_lev0, nlev, scale, flags, f, fs, x, xs, y, ys, alpha);

    const ElemSizes& es = c->es;
    ElemArgs a;
    a.wfin = nullptr; a.wsslot = nullptr; a.wcnt = nullptr; a.wfence = 0;
    a.nEl = c->nEl; a.nlev = nlev; a.lev0 = geom_lev0; a.total = c->nEl*nlev;
    a.flags = flags; a.scale = scale; a.alpha = alpha;
    a.J = c->d_J; a.det = c->d_det; a.tI = c->d_tI; a.th = c->d_th; a.tIp = c->d_tIp; a.tnp = c->nk/2 + 1; a.tps = (long long)c->nEl*c->es.mp12*2; a.tnode = 0; a.E = c->d_E; a.w = c->d_w;
    a.i0 = c->d_i0; a.i1x = c->d_i1x; a.i1y = c->d_i1y; a.i2 = c->d_i2; a.iq = c->d_iq;
    if (in == 3 && !c->d_iq) return MIMSEM_ERR_STATE;     // projection operators need mimsem_mesh_desc::indsq
    a.f = f; a.fs = fs; a.x = x; a.xs = xs;
    a.f2 = f2; a.f2s = f2s; a.param = param; a.xn = c->d_xn;
    {
        a.lch = level_chunk(c, nlev);
        a.swz = 0;   // pass 1: the natural order already keeps all level-chunks of an element on one XCD (profiles/r01_swizzle_ab.txt)
    }
    int rc;
    c->ev_k1[0] = c->ev_k1[1] = c->ev_k2[0] = c->ev_k2[1] = nullptr;
    if (c->profiling && (c->prof_count++ % c->prof_every) == 0) { c->ev_k1[0] = c->next_event(); c->ev_k1[1] = c->next_event(); c->ev_k2[0] = c->next_event(); c->ev_k2[1] = c->next_event(); c->ev_has2.push_back(0); }
    if (outsp == 2) {
        a.out = y; a.os = ys;
        // Wmat itself stays on k_elem_apply by default: measured 9.3e9 applies/s there against 8.2e9 on the DPP kernel (its element algebra
        // is four DPP stages for 9 values); Whmat +3 %, WtQUmat +14 % (profiles/r02_wave_ab.txt).  MIMSEM_WAVE2=2 includes Wmat, 0 none.
        const bool wave2_op = (op == MIMSEM_OP_WMAT && c->wave2_mode == 2) || op == MIMSEM_OP_WHMAT || op == MIMSEM_OP_WTQUMAT || op == MIMSEM_OP_WTQDUDZ;
        const bool fits = (long long)c->n1 < (1LL << 28) && (long long)c->n2 < (1LL << 28) && (long long)c->nEl*es.mp12 < (1LL << 28);
        if (c->wave1 && es.n == 3 && wave2_op && fits && part == 0 && c->wave2_mode != 0) {
            // p = 3: the wave-level kernel of the 2-form-valued operators (no scatter: one launch)
            if ((rc = c->ensure_ye(64))) return rc;
            a.wlane = c->d_wlane; a.wplan = nullptr; a.wgroups = c->w_ngroups; a.wg0 = 0; a.wdump = 0; a.wsing = nullptr; a.wnode = nullptr;
            a.wG = c->d_wG; a.wR = c->d_wR; a.wstamps = nullptr;
            a.lch = wave_level_chunk(c, nlev); a.wcpp = wave_chunks_per_item(c, nlev, a.lch, c->w_ngroups); a.swz = c->wave_order;
            a.accum = (flags & MIMSEM_FLAG_ACCUM) ? 1 : 0; a.y = c->d_ye;       // a.y: the dump for idle lanes
            for (size_t k = 0; k < 20; k++) a.Etab[k] = k < c->tab.E.size() ? c->tab.E[k] : 0.0;
            for (size_t k = 0; k < 5; k++) a.Wq[k] = k < c->tab.quad.w.size() ? c->tab.quad.w[k] : 0.0;
            rc = launch_apply_wave2(c, op, a);
            c->ev_k1[0] = c->ev_k1[1] = c->ev_k2[0] = c->ev_k2[1] = nullptr;
            return rc;
        }
        rc = launch_elem_apply(c, op, a);
        c->ev_k1[0] = c->ev_k1[1] = c->ev_k2[0] = c->ev_k2[1] = nullptr;
        return rc;
    }
    a.fperm = nullptr; a.accum = (flags & MIMSEM_FLAG_ACCUM) ? 1 : 0;
    a.d0 = a.d1x = a.d1y = nullptr; a.y = nullptr; a.ys = 0;
    if (epi) {
        // Richardson sweep: element pass into the workspace, then the gather with the update epilogue (y is the iterate);
        // with `blocks` the element-block preconditioner runs in between on the second half of the workspace
        const long long per = (long long)c->nEl*(outsp == 1 ? 2*es.n1e : es.n0e);
        if ((rc = c->ensure_ye(per*nlev*(blocks ? 2 : 1)))) return rc;
        a.out = c->d_ye; a.os = per;
        a.flags = flags & ~MIMSEM_FLAG_ACCUM;
        if ((rc = launch_elem_apply(c, op, a))) return rc;
        const double* src = c->d_ye;
        if (blocks) {
            double* ze = c->d_ye + per*nlev;
            if ((rc = launch_blocks_residual(c, nlev, blocks, c->d_ye, per, epi->b, epi->bs, ze, per, epi->escale, epi->ess))) return rc;
            src = ze;
        }
        return launch_gather_epilogue(c, outsp, nlev, src, per, *epi, y, ys);
    }
    if (outsp == 1 && c->fused1 && op < MIMSEM_OP_UMAT_UP && op != MIMSEM_OP_UMAT_RAY) {
        // fused path: group-local sums in LDS, complete slots written straight to y, perimeter partials to the workspace
        if ((rc = c->ensure_ye((long long)std::max(c->f_npart, 1)*nlev))) return rc;
        a.fperm = c->d_fperm; a.flid = c->d_flid; a.fslot = c->d_fslot; a.fcnt = c->d_fcnt;
        a.ngroups = c->f_ngroups; a.lmax = c->f_lmax;
        a.y = y; a.ys = ys; a.out = c->d_ye; a.os = c->f_npart;
        a.flags = flags & ~MIMSEM_FLAG_ACCUM;
        rc = launch_elem_apply(c, op, a);
        if (!rc) rc = launch_gather_perim(c, nlev, c->d_ye, c->f_npart, a.accum, y, ys);
        c->ev_k1[0] = c->ev_k1[1] = c->ev_k2[0] = c->ev_k2[1] = nullptr;
        return rc;
    }
    // the wave kernel addresses its rows with 32-bit byte offsets: vectors / metric beyond 4 GB per level stay on the two-pass form
    const bool wave_fits = (long long)c->n1 < (1LL << 28) && (long long)c->n0 < (1LL << 28) && (long long)c->n2 < (1LL << 28) &&
                           (long long)c->nEl*es.mp12 < (1LL << 28);
    if (c->wave1 && wave_fits && wave_op) {
        // wave-level fused path: complete slots straight into y, one partial per perimeter slot into the workspace, perimeter pass
        const long long prow = (long long)c->w_npart + 128;              // partial sums of a level + the dump tail (64 lanes x 16 bytes)
        int g0 = 0, g1 = c->w_ngroups, r0 = 0, r1 = c->w_nps;
        if (splits && part == MIMSEM_PART_BOUNDARY) { g1 = c->w_nbgroups; r1 = c->w_nbrec; }
        if (splits && part == MIMSEM_PART_INTERIOR) { g0 = c->w_nbgroups; r0 = c->w_nbrec; }
        a.wlane = c->d_wlane; a.wplan = c->d_wplan; a.wgroups = g1 - g0; a.wg0 = g0; a.wdump = c->w_npart;
        a.wsing = c->w_nsing ? c->d_wsing : nullptr; a.wnode = c->d_wnode; a.wG = c->d_wG; a.wR = c->d_wR;
        if (c->d_tIn) { a.tIp = c->d_tIn; a.tps = (long long)c->n0*2; a.tnode = 1; }     // (MIMSEM_WAVE_TNODE=1: thickInv per node)
        a.lch = wave_level_chunk(c, nlev);
        a.wcpp = wave_chunks_per_item(c, nlev, a.lch, g1 - g0);
        a.swz = c->wave_order;
        a.wtfin = nullptr; a.wtile = 0;
        if (c->w_ntiles > 0) {                                           // tile mode (never together with a halo split: setup_wave)
            if (splits) return MIMSEM_ERR_STATE;
            a.wtfin = c->d_wtfin; a.wtile = WTF;
            a.wcpp = std::min(a.wcpp, MIMSEM_WTLEV/8);                   // the tile's LDS rows hold MIMSEM_WTLEV levels
            a.swz &= ~2;                                                 // group-minor work items: the four waves of a workgroup = the four groups of a tile
        }
        // the whole operator in ONE launch: each side of the perimeter is finished by the group that reaches it second (not for the
        // parts of a split apply: their partial sums wait for the other part, and the perimeter pass finishes them)
        const bool fin = c->w_fin && !splits && !a.wtfin && (nlev + a.lch*a.wcpp - 1)/(a.lch*a.wcpp) <= std::max(c->nk, 1);
        a.wfin = fin ? c->d_wfin : nullptr; a.wsslot = c->d_wsslot; a.wcnt = c->d_wcnt;
        a.wfence = (fin && c->w_partmem == 2 && exp_env("MIMSEM_WAVE_FIN_FENCE") && atoi(exp_env("MIMSEM_WAVE_FIN_FENCE")) != 0) ? 1 : 0;
        if (splits) {
            // the pending BOUNDARY part and its INTERIOR part must match; nothing else can consume or overwrite the partial sums
            if (part == MIMSEM_PART_BOUNDARY) {
                if (c->split.pending) return MIMSEM_ERR_STATE;
            } else if (!c->split.pending || c->split.op != op || c->split.lev0 != geom_lev0 || c->split.nlev != nlev ||
                       c->split.flags != flags || c->split.y != y || c->split.ys != ys) return MIMSEM_ERR_STATE;
        }
        if ((rc = fin ? c->ensure_wpart(prow*nlev) : (splits ? c->ensure_wsplit(prow*nlev) : c->ensure_ye(prow*nlev)))) return rc;
        double* const prt = fin ? c->d_wpart : (splits ? c->d_wsplit : c->d_ye);
        a.y = y; a.ys = ys; a.out = prt; a.os = prow;
        a.flags = flags & ~MIMSEM_FLAG_ACCUM;
        a.wstamps = nullptr;
        for (size_t k = 0; k < 20; k++) a.Etab[k] = k < c->tab.E.size() ? c->tab.E[k] : 0.0;
#ifdef MIMSEM_STAMPS      // diagnostic build: per-phase s_memtime stamps of every work item of this launch, summarised on stderr
        static long long* d_st = nullptr; static size_t st_items = 0;
        const size_t items = (size_t)c->w_ngroups*((nlev + a.lch - 1)/a.lch);
        if (exp_env("MIMSEM_WAVE_STAMPS")) {
            if (items > st_items) { if (d_st) (void)hipFree(d_st); (void)hipMalloc((void**)&d_st, items*16*8); st_items = items; }
            (void)hipMemsetAsync(d_st, 0, items*16*8, c->stream);
            a.wstamps = d_st;
        }
#endif
        rc = a.wgroups > 0 ? launch_apply_wave(c, op, a) : MIMSEM_OK;
#ifdef MIMSEM_STAMPS
        if (a.wstamps) {
            (void)hipStreamSynchronize(c->stream);
            std::vector<long long> h(items*16);
            (void)hipMemcpy(h.data(), d_st, items*16*8, hipMemcpyDeviceToHost);
            long long t0 = h[0], t1 = 0;
            for (size_t i = 0; i < items; i++) { t0 = std::min(t0, h[i*16]); t1 = std::max(t1, h[i*16 + 15]); }
            fprintf(stderr, "[stamps] items %zu lch %d  first entry -> last done: %lld ticks\n", items, a.lch, t1 - t0);
            const char* names[16] = {"entry(rel. first)", "barrier", "tables requested", "level loads issued", "batch0", "batch1", "batch2", "batch3",
                                     "b4", "b5", "b6", "chunks done", "fin: stores acked", "fin: arrival counted", "fin: sides finished", "stores acked"};
            for (int k = 0; k < 16; k++) {
                std::vector<long long> v;
                for (size_t i = 0; i < items; i++) if (h[i*16 + k]) v.push_back(k == 0 ? h[i*16] - t0 : h[i*16 + k] - h[i*16]);
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                fprintf(stderr, "[stamps] %-20s min %8lld  p10 %8lld  median %8lld  p90 %8lld  max %8lld   (ticks since the wave's entry)\n", names[k],
                        v.front(), v[v.size()/10], v[v.size()/2], v[v.size()*9/10], v.back());
            }
        }
#endif
        if (!rc && !fin) rc = launch_wave_perim(c, nlev, prt, prow, a.accum, y, ys, r0, r1);
        if (splits && !rc) {
            if (part == MIMSEM_PART_BOUNDARY) { c->split.pending = true; c->split.op = op; c->split.lev0 = geom_lev0; c->split.nlev = nlev;
                                                c->split.flags = flags; c->split.y = y; c->split.ys = ys; }
            else c->split.pending = false;
        }
        c->ev_k1[0] = c->ev_k1[1] = c->ev_k2[0] = c->ev_k2[1] = nullptr;
        return rc;
    }
    const long long per = (long long)c->nEl*(outsp == 1 ? 2*es.n1e : es.n0e);
    if ((rc = c->ensure_ye(per*nlev))) return rc;
    a.out = c->d_ye; a.os = per;
    a.flags = flags & ~MIMSEM_FLAG_ACCUM;
    if (c->direct) { a.d0 = c->d_d0; a.d1x = c->d_d1x; a.d1y = c->d_d1y; a.y = y; a.ys = ys; a.accum = (flags & MIMSEM_FLAG_ACCUM) ? 1 : 0; }
    rc = launch_elem_apply(c, op, a);
    if (!rc) rc = launch_gather_sum(c, outsp, nlev, c->d_ye, per, (flags & MIMSEM_FLAG_ACCUM) ? 1 : 0, y, ys, c->direct);
    c->ev_k1[0] = c->ev_k1[1] = c->ev_k2[0] = c->ev_k2[1] = nullptr;
    return rc;
}

extern "C" {

int mimsem_op_elmat_size(const mimsem_ctx* c, int op) {
    if (!c) return MIMSEM_ERR_ARG;
    const ElemSizes& es = c->es;
    switch (op) {
    case MIMSEM_OP_UMAT: case MIMSEM_OP_UHMAT: case MIMSEM_OP_UTMAT: case MIMSEM_OP_UTMAT_H: case MIMSEM_OP_UMAT_RAY:
        return 4*es.n1e*es.n1e;
    case MIMSEM_OP_ROTMAT: return 2*es.n1e*es.n1e;
    case MIMSEM_OP_WMAT: case MIMSEM_OP_WHMAT: case MIMSEM_OP_WMATINV: case MIMSEM_OP_WHMATINV: return es.n2e*es.n2e;
    case MIMSEM_OP_PMAT: case MIMSEM_OP_PHMAT: return es.n0e*es.n0e;
    case MIMSEM_OP_WTQUMAT: case MIMSEM_OP_WTQDUDZ: case MIMSEM_OP_UTQWMAT: return 2*es.n2e*es.n1e;
    }
    return MIMSEM_ERR_ARG;
}

int mimsem_op_element_matrices_ex(mimsem_ctx* c, int op, int geom_lev, double scale, double tau, unsigned flags,
                                  const double* f, const double* u, double* out) {
    if (!c || !out || !f || !u) return MIMSEM_ERR_ARG;
    if (op != MIMSEM_OP_UMAT_RAY) return MIMSEM_ERR_ARG;
    if (geom_lev < 0 || geom_lev >= c->nk) return MIMSEM_ERR_ARG;
    return launch_elmats(c, op, geom_lev, scale, flags, f, out, u, tau);
}

int mimsem_op_element_matrices(mimsem_ctx* c, int op, int geom_lev, double scale, unsigned flags,
                               const double* f, double* out) {
    if (!c || !out) return MIMSEM_ERR_ARG;
    if (is_up_op(op)) return MIMSEM_ERR_ARG;
    int in, cf, outsp;
    if (op_spaces(op, &in, &cf, &outsp)) return MIMSEM_ERR_ARG;
    if (cf >= 0 && !f) return MIMSEM_ERR_ARG;
    if (geom_lev < 0 || geom_lev >= c->nk) return MIMSEM_ERR_ARG;
    if (op == MIMSEM_OP_UTMAT && geom_lev + 1 >= c->nk) return MIMSEM_ERR_ARG;
    if (op == MIMSEM_OP_WMATINV || op == MIMSEM_OP_WHMATINV) {
        // Wmat/Whmat blocks with the inverse ops' fixed flags, then batched Gauss-Jordan (LinAlg.cpp:186-269)
        const int base = (op == MIMSEM_OP_WMATINV) ? MIMSEM_OP_WMAT : MIMSEM_OP_WHMAT;
        int rc = launch_elmats(c, base, geom_lev, scale, MIMSEM_FLAG_VERT, f, out);
        if (rc) return rc;
        return mimsem_block_inverse_inplace(c, c->nEl, c->es.n2e, out);
    }
    return launch_elmats(c, op, geom_lev, scale, flags, f, out);
}

int mimsem_block_inverse(mimsem_ctx* c, long long nblocks, int n, double* blocks) {
    if (!c || nblocks < 0 || n < 1 || (nblocks && !blocks)) return MIMSEM_ERR_ARG;
    return mimsem_block_inverse_inplace(c, nblocks, n, blocks);      // MIMSEM_ERR_UNSUPPORTED when one block no longer fits a workgroup's LDS
}

// the same, and the number of blocks in which LinAlg::Inv would have returned its error (a pivot below 1e-12 after full pivoting,
// eul/LinAlg.cpp:243-246): MIMSEM_OK with *n_singular > 0 -- the blocks are inverted as the reference inverts them, garbage included
int mimsem_block_inverse_status(mimsem_ctx* c, long long nblocks, int n, double* blocks, int* n_singular) {
    if (!c || !n_singular || nblocks < 0 || n < 1 || (nblocks && !blocks)) return MIMSEM_ERR_ARG;
    *n_singular = 0;
    if (c->is_capturing()) return MIMSEM_ERR_STATE;                  // (reads a counter back)
    int* d_err = nullptr;
    MIMSEM_HIP_TRY(hipMalloc((void**)&d_err, sizeof(int)));
    int rc = MIMSEM_OK;
    if (hipMemsetAsync(d_err, 0, sizeof(int), c->stream) != hipSuccess) rc = MIMSEM_ERR_HIP;
    if (!rc) rc = mimsem_block_inverse_inplace(c, nblocks, n, blocks, d_err);
    if (!rc && hipMemcpyAsync(n_singular, d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = MIMSEM_ERR_HIP;
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = MIMSEM_ERR_HIP;
    (void)hipFree(d_err);
    return rc;
}

int mimsem_elem_blocks_apply(mimsem_ctx* c, int form, int nlev, unsigned flags, const double* blocks, long long blocks_level_stride,
                             const double* elem_scale, long long elem_scale_stride,
                             const double* x, long long xs, double* y, long long ys, double alpha) {
    if (!c || !blocks || !x || !y || form < 0 || form > 2 || nlev < 0) return MIMSEM_ERR_ARG;
    return launch_blocks_apply(c, form, nlev, (flags & MIMSEM_FLAG_TRANSPOSE) ? 1 : 0, blocks, blocks_level_stride,
                               x, xs, y, ys, alpha, (flags & MIMSEM_FLAG_ACCUM) ? 1 : 0, elem_scale, elem_scale_stride);
}

int mimsem_incidence_apply(mimsem_ctx* c, int which, int nlev, const double* x, long long xs, double* y, long long ys) {
    if (!c || !x || !y || which < 0 || which > 3 || nlev < 0) return MIMSEM_ERR_ARG;
    return launch_incidence(c, which, nlev, x, xs, y, ys);
}

int mimsem_interp_quad(mimsem_ctx* c, int form, unsigned flags, int nlev, const double* x, long long xs, double* out, long long os) {
    if (!c || form < 0 || form > 2 || nlev < 0 || (flags & ~MIMSEM_INTERP_GLOBAL)) return MIMSEM_ERR_ARG;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (!x || !out) return MIMSEM_ERR_ARG;
    const long long per = (long long)c->nEl*c->es.mp12*(form == 1 ? 2 : 1);
    if (nlev > 1 && os < per) return MIMSEM_ERR_ARG;
    return launch_interp_quad(c, form, (flags & MIMSEM_INTERP_GLOBAL) ? 1 : 0, nlev, x, xs, out, os);
}

int mimsem_sw_operator_apply(mimsem_ctx* c, int nlev, double a, double grav, double H, const double* f0, long long f0s,
                             const double* x, long long xs, double* y, long long ys) {
    if (!c || nlev < 0) return MIMSEM_ERR_ARG;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (!f0 || !x || !y || x == y) return MIMSEM_ERR_ARG;
    const long long n = (long long)c->n1 + c->n2;
    if (nlev > 1 && (xs < n || ys < n)) return MIMSEM_ERR_ARG;
    return launch_sw_operator(c, nlev, a, grav, H, f0, f0s, x, xs, y, ys);
}

int mimsem_sw_blocks_apply(mimsem_ctx* c, int nlev, const double* blocks, const double* x, long long xs, double* y, long long ys) {
    if (!c || nlev < 0) return MIMSEM_ERR_ARG;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (!blocks || !x || !y || x == y) return MIMSEM_ERR_ARG;
    const long long n = (long long)c->n1 + c->n2;
    if (nlev > 1 && (xs < n || ys < n)) return MIMSEM_ERR_ARG;
    return launch_sw_blocks_apply(c, nlev, blocks, x, xs, y, ys);
}

int mimsem_op_richardson_sweep(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, double tau, unsigned flags,
                               const double* f, long long fs, const double* u, long long us,
                               const double* b, long long bs, const double* dinv, long long ds,
                               double* x, long long xs, double* upd, long long upds) {
    if (!c || !b || !dinv || !x || (flags & MIMSEM_FLAG_ACCUM)) return MIMSEM_ERR_ARG;
    int in, cf, outsp;
    if (op_spaces(op, &in, &cf, &outsp) || outsp == 2 || in != outsp) return MIMSEM_ERR_ARG;     // square operators on gathered spaces
    if (is_up_op(op) && !u) return MIMSEM_ERR_ARG;
    GatherEpilogue g{1, b, bs, dinv, ds, upd, upds};
    return op_apply_core(c, op, geom_lev0, nlev, scale, flags, f, fs, u, us, tau, x, xs, x, xs, 1.0, &g);
}

// the Chebyshev form of mimsem_op_richardson_sweep: z = dinv (b - Op x);  p = z + beta p;  x += alpha p  (two launches)
int mimsem_op_chebyshev_sweep(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, double tau, unsigned flags,
                              const double* f, long long fs, const double* u, long long us,
                              const double* b, long long bs, const double* dinv, long long ds, double alpha, double beta,
                              double* p, long long ps, double* x, long long xs, double* upd, long long upds) {
    if (!c || !b || !dinv || !x || !p || (flags & MIMSEM_FLAG_ACCUM)) return MIMSEM_ERR_ARG;
    int in, cf, outsp;
    if (op_spaces(op, &in, &cf, &outsp) || outsp == 2 || in != outsp) return MIMSEM_ERR_ARG;     // square operators on gathered spaces
    if (is_up_op(op) && !u) return MIMSEM_ERR_ARG;
    GatherEpilogue g{5, b, bs, dinv, ds, upd, upds};
    g.alpha = alpha; g.beta = beta; g.p = p; g.ps = ps;
    return op_apply_core(c, op, geom_lev0, nlev, scale, flags, f, fs, u, us, tau, x, xs, x, xs, 1.0, &g);
}

int mimsem_block_richardson_sweep(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                                  const double* f, long long fs, const double* blocks,
                                  const double* b, long long bs, double* x, long long xs, double* upd, long long upds) {
    if (!c || !b || !blocks || !x || (flags & MIMSEM_FLAG_ACCUM)) return MIMSEM_ERR_ARG;
    int in, cf, outsp;
    if (op_spaces(op, &in, &cf, &outsp) || outsp != 1 || in != 1 || is_up_op(op)) return MIMSEM_ERR_ARG;
    if (c->es.n > 5) return MIMSEM_ERR_UNSUPPORTED;
    GatherEpilogue g{2, b, bs, nullptr, 0, upd, upds};
    return op_apply_core(c, op, geom_lev0, nlev, scale, flags, f, fs, nullptr, 0, 0.0, x, xs, x, xs, 1.0, &g, blocks);
}

int mimsem_block_chebyshev_sweep(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                                 const double* f, long long fs, const double* blocks, const double* elem_scale, long long es_stride,
                                 const double* b, long long bs, double alpha, double beta, double* p, long long ps,
                                 double* x, long long xs, double* upd, long long upds) {
    if (!c || !b || !blocks || !x || !p || (flags & MIMSEM_FLAG_ACCUM)) return MIMSEM_ERR_ARG;
    int in, cf, outsp;
    if (op_spaces(op, &in, &cf, &outsp) || outsp != 1 || in != 1 || is_up_op(op)) return MIMSEM_ERR_ARG;
    if (c->es.n > 5) return MIMSEM_ERR_UNSUPPORTED;
    GatherEpilogue g{3, b, bs, nullptr, 0, upd, upds};
    g.alpha = alpha; g.beta = beta; g.p = p; g.ps = ps; g.escale = elem_scale; g.ess = es_stride;
    return op_apply_core(c, op, geom_lev0, nlev, scale, flags, f, fs, nullptr, 0, 0.0, x, xs, x, xs, 1.0, &g, blocks);
}

// A whole fixed-length Chebyshev solve of  Umat x = b  from x = 0 in ONE call (round 6): the first step has no element pass (Op 0 = 0: the block
// pass takes b as its residual) and writes x and p instead of updating them, so neither needs clearing; steps 1 .. nsteps - 1 are the three
// launches of mimsem_block_chebyshev_sweep.  The same bits as nsteps calls of that entry on x = 0, p = anything finite (up to the sign of a zero).
// EXPERIMENT (experiments build, MIMSEM_CHEB_PEND=1): two launches per step -- the gather epilogue of step k folded into the element pass of
// step k + 1 (body_elem_apply<..., PEND>), x and p alternating between two buffers.  Bit-equal and SLOWER (8.22 against 7.07 ms per HorizSolve
// evaluation): the folded update turns the epilogue's coalesced slot-order accesses into 96 gathers per unit on 12 of 32 lanes.
int mimsem_block_chebyshev_solve(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                                 const double* f, long long fs, const double* blocks, const double* elem_scale, long long es_stride,
                                 const double* b, long long bs, int nsteps, const double* coef,
                                 double* x, long long xs, double* pb, long long pbs, double* upd, long long upds) {
    if (!c || nlev < 0 || nsteps < 1 || !coef || (flags & ~MIMSEM_FLAG_VERT)) return MIMSEM_ERR_ARG;
    if (op != MIMSEM_OP_UMAT) return MIMSEM_ERR_UNSUPPORTED;
    if (c->es.n > 5) return MIMSEM_ERR_UNSUPPORTED;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (!b || !blocks || !x || x == b || geom_lev0 < 0 || geom_lev0 + nlev > c->nk) return MIMSEM_ERR_ARG;
    if (nlev > 1 && (xs < c->n1 || bs < c->n1 || (pb && pbs < c->n1) || (upd && upds < c->n1))) return MIMSEM_ERR_ARG;
    (void)f; (void)fs;
    const ElemSizes& es = c->es;
    const long long per = (long long)c->nEl*2*es.n1e, n1 = c->n1;
    const bool pend = kExperiments && c->cheb_pend && nsteps > 1;
    int rc;
    if ((rc = c->ensure_ye(per*nlev*2))) return rc;
    if ((rc = c->ensure_cheb((pend ? 3 : 1)*n1*nlev))) return rc;
    double* ye = c->d_ye; double* ze = c->d_ye + per*nlev;
    ElemArgs a;
    a.wfin = nullptr; a.wsslot = nullptr; a.wcnt = nullptr; a.wfence = 0;
    a.nEl = c->nEl; a.nlev = nlev; a.lev0 = geom_lev0; a.total = c->nEl*nlev;
    a.flags = flags; a.scale = scale; a.alpha = 1.0;
    a.J = c->d_J; a.det = c->d_det; a.tI = c->d_tI; a.th = c->d_th; a.tIp = c->d_tIp; a.tnp = c->nk/2 + 1; a.tps = (long long)c->nEl*es.mp12*2; a.tnode = 0; a.E = c->d_E; a.w = c->d_w;
    a.i0 = c->d_i0; a.i1x = c->d_i1x; a.i1y = c->d_i1y; a.i2 = c->d_i2; a.iq = c->d_iq;
    a.f = nullptr; a.fs = 0; a.f2 = nullptr; a.f2s = 0; a.param = 0.0; a.xn = c->d_xn;
    a.lch = level_chunk(c, nlev); a.swz = 0;
    a.fperm = nullptr; a.accum = 0; a.d0 = a.d1x = a.d1y = nullptr; a.y = nullptr; a.ys = 0;
    a.out = ye; a.os = per;
    c->ev_k1[0] = c->ev_k1[1] = c->ev_k2[0] = c->ev_k2[1] = nullptr;
    // step 0: z_0 = P b
    if ((rc = launch_blocks_residual(c, nlev, blocks, nullptr, per, b, bs, ze, per, elem_scale, es_stride))) return rc;
    GatherEpilogue g{3, b, bs, nullptr, 0, nullptr, 0};
    if (!pend) {
        double* p = c->d_cheb;
        g.p = p; g.ps = n1;
        for (int k = 0; k < nsteps; k++) {
            if (k > 0) {
                a.x = x; a.xs = xs;
                if ((rc = launch_elem_apply(c, MIMSEM_OP_UMAT, a))) return rc;
                if ((rc = launch_blocks_residual(c, nlev, blocks, ye, per, b, bs, ze, per, elem_scale, es_stride))) return rc;
            }
            g.alpha = coef[2*k]; g.beta = coef[2*k + 1]; g.zero = k == 0;
            // the check vectors: pb = z_0, upd = z_{nsteps-1}; a one-step solve has one z for both
            g.upd = k == nsteps - 1 && upd ? upd : (k == 0 ? pb : nullptr); g.us = k == nsteps - 1 && upd ? upds : pbs;
            if ((rc = launch_gather_epilogue(c, 1, nlev, ze, per, g, x, xs))) return rc;
        }
        if (nsteps == 1 && upd && pb)
            for (int l = 0; l < nlev; l++) MIMSEM_HIP_TRY(hipMemcpyAsync(pb + (size_t)l*pbs, upd + (size_t)l*upds, (size_t)n1*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return MIMSEM_OK;
    }
    // after nsteps - 1 alternations the iterate must sit in the caller's x
    double* X[2]; long long XS[2];
    const int cx = (nsteps - 1) & 1;
    X[cx] = x; XS[cx] = xs; X[cx ^ 1] = c->d_cheb; XS[cx ^ 1] = n1;
    double* P[2] = {c->d_cheb + n1*nlev, c->d_cheb + 2*n1*nlev};
    int cur = 0;
    for (int k = 1; k < nsteps; k++) {
        ElemPending pd;
        pd.plan = c->d_g1; pd.ze = ze; pd.zes = per; pd.alpha = coef[2*(k - 1)]; pd.beta = coef[2*(k - 1) + 1]; pd.first = k == 1;
        pd.p_in = P[cur]; pd.p_out = P[cur ^ 1]; pd.ps = n1;
        pd.x_out = X[cur ^ 1]; pd.xos = XS[cur ^ 1];
        pd.upd = k == 1 ? pb : nullptr; pd.us = pbs;
        a.x = X[cur]; a.xs = XS[cur];
        if ((rc = launch_elem_apply_pending(c, a, pd))) return rc;
        cur ^= 1;
        if ((rc = launch_blocks_residual(c, nlev, blocks, ye, per, b, bs, ze, per, elem_scale, es_stride))) return rc;
    }
    g.upd = upd; g.us = upds;
    g.alpha = coef[2*(nsteps - 1)]; g.beta = coef[2*(nsteps - 1) + 1]; g.p = P[cur]; g.ps = n1;
    return launch_gather_epilogue(c, 1, nlev, ze, per, g, X[cur], XS[cur]);
}

int mimsem_sw_operator_precond_apply(mimsem_ctx* c, int nlev, double a, double grav, double H, const double* f0, long long f0s,
                                     const double* blocks, const double* x, long long xs, double* z, long long zs) {
    if (!c || nlev < 0) return MIMSEM_ERR_ARG;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (!f0 || !blocks || !x || !z || x == z) return MIMSEM_ERR_ARG;
    const long long n = (long long)c->n1 + c->n2;
    if (nlev > 1 && (xs < n || zs < n)) return MIMSEM_ERR_ARG;
    return launch_sw_operator_precond(c, nlev, a, grav, H, f0, f0s, blocks, x, xs, z, zs);
}

int mimsem_sw_operator_precond_chebyshev(mimsem_ctx* c, int nlev, double a, double grav, double H, const double* f0, long long f0s,
                                         const double* blocks, double ca, double cb, double* x, long long xs, double* r, long long rs,
                                         double* d, long long ds) {
    if (!c || nlev < 0) return MIMSEM_ERR_ARG;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (!f0 || !blocks || !x || !r || !d || x == r || x == d || r == d) return MIMSEM_ERR_ARG;
    const long long n = (long long)c->n1 + c->n2;
    if (nlev > 1 && (xs < n || rs < n || ds < n)) return MIMSEM_ERR_ARG;
    return launch_sw_operator_precond_chebyshev(c, nlev, a, grav, H, f0, f0s, blocks, ca, cb, x, xs, r, rs, d, ds);
}

int mimsem_sw_chebyshev_step2(mimsem_ctx* c, int nlev, double a, double grav, double H, const double* f0, long long f0s, const double* blocks,
                              int pending, double pca, double pcb, double ca, double cb, double* x, long long xs,
                              const double* r_in, const double* d_in, double* r_out, double* d_out, double* rh, double* dh, long long vs) {
    if (!c || nlev < 0) return MIMSEM_ERR_ARG;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (!f0 || !blocks || !x || !r_in || !d_in || !rh || !dh) return MIMSEM_ERR_ARG;
    if (pending && (!r_out || !d_out || r_out == r_in || d_out == d_in || r_out == d_in || d_out == r_in)) return MIMSEM_ERR_ARG;   // (the other elements of a slot read the old values in the same launch)
    const long long n = (long long)c->n1 + c->n2;
    if (nlev > 1 && (xs < n || vs < n)) return MIMSEM_ERR_ARG;
    return launch_sw_chebyshev_step2(c, nlev, a, grav, H, f0, f0s, blocks, pending, pca, pcb, ca, cb, x, xs, r_in, d_in, r_out, d_out, rh, dh, vs);
}
int mimsem_sw_chebyshev_flush(mimsem_ctx* c, int nlev, double ca, double cb, double* x, long long xs, double* r, double* d, long long vs) {
    if (!c || nlev < 0) return MIMSEM_ERR_ARG;
    if (nlev == 0 || c->nEl == 0) return MIMSEM_OK;
    if (!x || !r || !d || x == r || x == d || r == d) return MIMSEM_ERR_ARG;
    return launch_sw_chebyshev_flush(c, nlev, ca, cb, x, xs, r, d, vs);
}

// the element-pass arguments of a single-level sweep on the src/ flavour (scale 1, no thickness), as op_apply_core fills them for its epilogue path
static void sweep_elem_args(mimsem_ctx* c, ElemArgs& a, const double* f, const double* f2, double param, const double* x, double* out, long long os) {
    a.wfin = nullptr; a.wsslot = nullptr; a.wcnt = nullptr; a.wfence = 0;
    a.nEl = c->nEl; a.nlev = 1; a.lev0 = 0; a.total = c->nEl;
    a.flags = 0; a.scale = 1.0; a.alpha = 1.0;
    a.J = c->d_J; a.det = c->d_det; a.tI = c->d_tI; a.th = c->d_th; a.tIp = c->d_tIp; a.tnp = c->nk/2 + 1; a.tps = (long long)c->nEl*c->es.mp12*2; a.tnode = 0; a.E = c->d_E; a.w = c->d_w;
    a.i0 = c->d_i0; a.i1x = c->d_i1x; a.i1y = c->d_i1y; a.i2 = c->d_i2; a.iq = c->d_iq;
    a.f = f; a.fs = 0; a.x = x; a.xs = 0;
    a.f2 = f2; a.f2s = 0; a.param = param; a.xn = c->d_xn;
    a.lch = level_chunk(c, 1); a.swz = 0;
    a.fperm = nullptr; a.accum = 0; a.d0 = a.d1x = a.d1y = nullptr; a.y = nullptr; a.ys = 0;
    a.out = out; a.os = os;
}

// Two independent fixed-length Chebyshev solves of a shallow-water Picard iteration, both from x = 0, in SHARED launches (round 6;
// csrc/elem_kernels.hip: k_sw_pair): exactly the sequence
//     for k < nA: mimsem_block_chebyshev_sweep(ctx, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, NULL, 0, blocks1, NULL, 0, b1, 0, coefA[2k], coefA[2k+1], p1, 0, x1, 0, k == nA-1 ? upd1 : (k == 0 ? pb1 : NULL), 0)
//     for k < nB: mimsem_op_chebyshev_sweep(ctx, MIMSEM_OP_PHMAT_UP, 0, 1, 1.0, tau, 0, h, 0, u, 0, b0, 0, dinv, 0, coefB[2k], coefB[2k+1], p0, 0, x0, 0, k == nB-1 ? upd0 : (k == 0 ? pb0 : NULL), 0)
// on x1 = x0 = 0 -- the same kernels' bodies, the same bits (up to the sign of a zero) -- with launch k of the first chain and launch k of the second in
// one grid.  Late in round 6 the chains lost what a zero start makes superfluous: no operator pass in either first step (the block pass takes b1 as its
// residual, the first q update is dinv b0), x and p WRITTEN there instead of updated (nothing to clear), and the first steps' preconditioned
// residuals -- P b1 and dinv b0, the reference norms of the checks -- come out as pb1 / pb0 instead of being computed again by the caller:
// 3 nA - 1 and 2 nB - 1 launches, 8 launches and fills fewer per Picard iteration around them.
int mimsem_sw_dual_chebyshev(mimsem_ctx* c, int nA, const double* coefA, const double* blocks1, const double* b1, double* p1, double* x1, double* upd1, double* pb1,
                             int nB, const double* coefB, double tau, const double* h, const double* u, const double* b0, const double* dinv,
                             double* p0, double* x0, double* upd0, double* pb0) {
    if (!c || nA < 1 || nB < 1 || !coefA || !coefB || !blocks1 || !b1 || !p1 || !x1 || !h || !u || !b0 || !dinv || !p0 || !x0) return MIMSEM_ERR_ARG;
    if ((pb1 && nA < 2 && upd1) || (pb0 && nB < 2 && upd0)) return MIMSEM_ERR_ARG;      // (a one-step solve has ONE preconditioned residual: ask for it once)
    if (c->nEl == 0) return MIMSEM_OK;
    const ElemSizes& es = c->es;
    if (es.n < 2 || es.n > 4) return MIMSEM_ERR_UNSUPPORTED;
    const long long per1 = (long long)c->nEl*2*es.n1e, per0 = (long long)c->nEl*es.n0e;
    int rc = c->ensure_ye(2*per1 + per0);
    if (rc) return rc;
    double *yeA = c->d_ye, *zeA = c->d_ye + per1, *yeQ = c->d_ye + 2*per1;
    ElemArgs ea, eq;
    sweep_elem_args(c, ea, nullptr, nullptr, 0.0, x1, yeA, per1);
    sweep_elem_args(c, eq, h, u, tau, x0, yeQ, per0);
    PairBlocks ba{c->nEl, 1, c->d_i1x, c->d_i1y, c->d_g1, blocks1, yeA, per1, b1, zeA, per1, (const int4*)c->d_bplan};
    PairGather ga{zeA, per1, c->d_g1, c->n1, GatherEpilogue{3, b1, 0, nullptr, 0, nullptr, 0}, x1};
    PairGather gq{yeQ, per0, c->d_g0, c->n0, GatherEpilogue{5, b0, 0, dinv, 0, nullptr, 0}, x0};
    ga.g.p = p1; ga.g.ps = 0; gq.g.p = p0; gq.g.ps = 0;
    c->ev_k1[0] = c->ev_k1[1] = c->ev_k2[0] = c->ev_k2[1] = nullptr;
    // chain A: {block pass on a zero result, epilogue that writes} then (nA - 1) x {element pass, block pass, epilogue};  chain q: {epilogue that
    // writes, no operator result} then (nB - 1) x {element pass, epilogue}
    const int LA = 3*nA - 1, LB = 2*nB - 1, L = std::max(LA, LB);
    for (int k = 0; k < L; k++) {
        int PA = -1, PB = -1;
        if (k < LA) {
            PA = k == 0 ? 3 : (k == 1 ? 2 : (k - 2)%3);
            if (PA == 2) {
                const int st = k == 1 ? 0 : 1 + (k - 2)/3;
                ga.g.alpha = coefA[2*st]; ga.g.beta = coefA[2*st + 1]; ga.g.zero = st == 0; ga.g.us = 0;
                ga.g.upd = st == nA - 1 && upd1 ? upd1 : (st == 0 ? pb1 : nullptr);
            }
        }
        if (k < LB) {
            PB = k == 0 ? 1 : (k - 1)%2;
            if (PB == 1) {
                const int st = k == 0 ? 0 : 1 + (k - 1)/2;
                gq.g.alpha = coefB[2*st]; gq.g.beta = coefB[2*st + 1]; gq.g.zero = st == 0; gq.g.noacc = st == 0; gq.g.us = 0;
                gq.g.upd = st == nB - 1 && upd0 ? upd0 : (st == 0 ? pb0 : nullptr);
            }
        }
        if (PA >= 0 && PB >= 0) rc = launch_sw_pair(c, PA, PB, ea, ba, ga, eq, gq);
        else if (PA == 0) rc = launch_elem_apply(c, MIMSEM_OP_UMAT, ea);                                  // (the longer chain's tail: its own kernels)
        else if (PA == 1) rc = launch_blocks_residual(c, 1, blocks1, yeA, per1, b1, 0, zeA, per1, nullptr, 0);
        else if (PA == 3) rc = launch_blocks_residual(c, 1, blocks1, nullptr, per1, b1, 0, zeA, per1, nullptr, 0);
        else if (PA == 2) rc = launch_gather_epilogue(c, 1, 1, zeA, per1, ga.g, x1, 0);
        else if (PB == 0) rc = launch_elem_apply(c, MIMSEM_OP_PHMAT_UP, eq);
        else rc = launch_gather_epilogue(c, 0, 1, yeQ, per0, gq.g, x0, 0);
        if (rc) return rc;
    }
    return MIMSEM_OK;
}

int mimsem_halo_pack(mimsem_ctx* c, const int* idx, int count, int nlev, const double* v, long long vs, double* buf) {
    if (!c || count < 0 || nlev < 0) return MIMSEM_ERR_ARG;
    if (count == 0 || nlev == 0) return MIMSEM_OK;        // empty message: pointers of empty arrays may be null
    if (!idx || !v || !buf) return MIMSEM_ERR_ARG;
    return launch_halo_pack(c, idx, count, nlev, v, vs, buf);
}
int mimsem_halo_unpack(mimsem_ctx* c, const int* idx, int count, int nlev, int mode, const double* buf, double* v, long long vs) {
    if (!c || count < 0 || nlev < 0) return MIMSEM_ERR_ARG;
    if (count == 0 || nlev == 0) return MIMSEM_OK;
    if (!idx || !v || !buf) return MIMSEM_ERR_ARG;
    return launch_halo_unpack(c, idx, count, nlev, mode, buf, v, vs);
}

// all neighbours in one launch; buf is segment-major [segment][level][slot]
int mimsem_halo_segments(mimsem_ctx* c, const int* idx, int nseg, const int* seg_off, int seg_begin, int seg_end, int nlev, int mode,
                         double* buf, double* v, long long vs) {
    if (!c || !idx || !seg_off || !buf || !v || nseg < 0 || nseg > MIMSEM_HALO_MAX_SEGMENTS || nlev < 0 || mode < 0 || mode > 2)
        return MIMSEM_ERR_ARG;
    if (seg_begin < 0 || seg_end < seg_begin || seg_end > nseg) return MIMSEM_ERR_ARG;
    for (int i = 0; i < nseg; i++) if (seg_off[i + 1] < seg_off[i]) return MIMSEM_ERR_ARG;
    return launch_halo_segments(c, idx, nseg, seg_off, seg_begin, seg_end, nlev, mode, buf, v, vs);
}

}  // extern "C"
