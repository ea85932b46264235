// cpm_hostmath.cpp -- see cpm_hostmath.h.  Standard C++ only.
//
// glm conventions the reference's host code computes with, kept so that the numbers agree (glm is not in the reference tree:
// unpinned): unit(v) = v * (1 / sqrt(v.v)); dot products summed left to right; mix(x, y, t) with a double t
// = x (1 - t) + y t evaluated in double; "differs from zero" = |c| >= epsilon.
#include "cpm_hostmath.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace cpm_host {

namespace {

inline Pt3 operator-(Pt3 a, Pt3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Pt3 operator+(Pt3 a, Pt3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Pt3 operator*(float s, Pt3 a) { return { a.x * s, a.y * s, a.z * s }; }
inline float inner(Pt3 a, Pt3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Pt3 unit(Pt3 a) { return (1.0f / std::sqrt(inner(a, a))) * a; }
inline Pt3 outer(Pt3 a, Pt3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }

inline Pt2 operator-(Pt2 a, Pt2 b) { return { a.x - b.x, a.y - b.y }; }
inline float inner(Pt2 a, Pt2 b) { return a.x * b.x + a.y * b.y; }

// > 0: q lies to the left of the directed line a -> b; 0: on it
inline float turn(Pt2 a, Pt2 b, Pt2 q) {
    const Pt2 ab = b - a, aq = q - a;
    return ab.x * aq.y - aq.x * ab.y;
}

// A counter-clockwise chain under construction on top of `cycle`; vertices below `base` are settled.
struct ChainBuilder {
    std::vector<Pt2>& cycle;
    size_t base;
    void extend(Pt2 q) {
        while (cycle.size() - base >= 2 && !(turn(cycle[cycle.size() - 2], cycle.back(), q) > 0.f)) cycle.pop_back();
        cycle.push_back(q);
    }
};

// closed interval that starts as {0}
struct Extent {
    float lo = 0.f, hi = 0.f;
    void cover(float t) { lo = std::min(lo, t); hi = std::max(hi, t); }
    float width() const { return hi - lo; }
};

}  // namespace

std::vector<Pt2> planeCoordinates(const std::vector<Pt3>& points, Pt3 through, Pt3 normal, Pt3 a0, Pt3 a1) {
    std::vector<Pt2> coords;
    coords.reserve(points.size());
    const float offset = inner(normal, through);
    for (Pt3 p : points) {
        const float height = inner(normal, p) - offset;     // signed distance from the plane
        const Pt3 inPlane = (p - height * normal) - through;  // foot of the perpendicular, seen from `through`
        coords.push_back({ inner(a0, inPlane), inner(a1, inPlane) });
    }
    return coords;
}

// Rule H -- what the rectangle search is handed.  With the points sorted by (x, y), L the run of points sharing the smallest
// x and R the run sharing the largest:
//   fewer than four points: the sorted points themselves;
//   one column (L is everything): bottom, top if it differs, bottom again;
//   otherwise a counter-clockwise walk that keeps strict left turns only --
//     1. from the first point of L along the points strictly right of the line (first of L -> last of R), ending at the last
//        point of R whatever side it is on;
//     2. if R has more than one point, the first point of R is appended;
//     3. on from there, back through the points (first of R down to the one behind L) strictly right of the line
//        (first of R -> last of L); turns are only undone down to the vertex step 2 ended with;
//     4. if L has more than one point, the last point of R is appended.
// For points in general position (no two sharing the extreme x) that is the convex hull, counter-clockwise from its
// lowest-leftmost vertex, collinear points dropped.  With tied columns -- an axis-parallel light: the cube's corners project
// in pairs -- steps 2 and 4 repeat vertices and step 3 never reaches the last point of L: the cycle then has chords among
// its edges and can miss that corner.  That is the reference's behaviour (lightcl/convexhull2d.cpp:84-127) and it decides
// which rectangle, and so which lattice point feeds which RNG stream; it is kept, as a rule, for that reason.
std::vector<Pt2> hullCycle(std::vector<Pt2> points) {
    std::sort(points.begin(), points.end(), [](Pt2 a, Pt2 b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    const size_t n = points.size();
    if (n < 4) return points;
    size_t lastOfL = 0;
    while (lastOfL + 1 < n && points[lastOfL + 1].x == points.front().x) ++lastOfL;
    const Pt2 bottomLeft = points.front(), topRight = points.back();
    if (lastOfL == n - 1) {
        std::vector<Pt2> column{ bottomLeft };
        if (topRight.y != bottomLeft.y) column.push_back(topRight);
        column.push_back(bottomLeft);
        return column;
    }
    size_t firstOfR = n - 1;
    while (firstOfR > 0 && !(topRight.x > points[firstOfR - 1].x)) --firstOfR;

    std::vector<Pt2> cycle{ bottomLeft };
    ChainBuilder under{ cycle, 0 };
    for (size_t i = lastOfL + 1; i + 1 < n; ++i)
        if (turn(bottomLeft, topRight, points[i]) < 0.f) under.extend(points[i]);
    under.extend(topRight);
    if (firstOfR != n - 1) cycle.push_back(points[firstOfR]);

    ChainBuilder over{ cycle, cycle.size() - 1 };
    const Pt2 bottomRight = points[firstOfR], topLeft = points[lastOfL];
    for (size_t i = firstOfR; i > lastOfL; --i)
        if (turn(bottomRight, topLeft, points[i]) < 0.f) over.extend(points[i]);
    if (lastOfL != 0) cycle.push_back(topRight);
    return cycle;
}

// Every edge (previous vertex -> vertex) of the cycle in turn, the closing edge first: the frame (along, across) anchored at the
// previous vertex, the extents of all vertices in it (each extent contains 0, the anchor), area = product of the widths.  A
// strictly smaller area replaces the best so far.  Repeated vertices have no direction (0 * inf) and are passed over.
Rectangle2 smallestRectangle(const std::vector<Pt2>& cycle) {
    Rectangle2 best;
    float bestArea = FLT_MAX;
    const size_t n = cycle.size();
    for (size_t head = 0; head < n; ++head) {
        const Pt2 anchor = cycle[(head + n - 1) % n];
        const Pt2 edge = cycle[head] - anchor;
        const float scale = 1.0f / std::sqrt(inner(edge, edge));
        const Pt2 along{ edge.x * scale, edge.y * scale };
        if (std::isnan(along.x) || std::isnan(along.y)) continue;
        const Pt2 across{ -along.y, along.x };
        Extent e0, e1;
        for (Pt2 q : cycle) {
            const Pt2 d = q - anchor;
            e0.cover(inner(d, along));
            e1.cover(inner(d, across));
        }
        const float area = e0.width() * e1.width();
        if (!(area < bestArea)) continue;
        bestArea = area;
        best.corner = { (anchor.x + e0.lo * along.x) + e1.lo * across.x, (anchor.y + e0.lo * along.y) + e1.lo * across.y };
        best.side0 = { along.x * e0.width(), along.y * e0.width() };
        best.side1 = { across.x * e1.width(), across.y * e1.width() };
    }
    return best;
}

LightRectangle fitLightRectangle(const std::vector<Pt3>& points, Pt3 through, Pt3 normal) {
    // first in-plane axis: the world axis (x if the normal leans more to x than to y, else y) dropped onto the plane, seen from
    // `through`; the second completes the right-handed frame (normal, a0, a1)
    const Pt3 worldAxis = std::fabs(normal.x) > std::fabs(normal.y) ? Pt3{ 1.f, 0.f, 0.f } : Pt3{ 0.f, 1.f, 0.f };
    const float offset = inner(normal, through);
    const Pt3 dropped = worldAxis - (inner(normal, worldAxis) - offset) * normal;
    const Pt3 a0 = unit(dropped - through);
    const Pt3 a1 = unit(outer(normal, a0));
    const Rectangle2 r = smallestRectangle(hullCycle(planeCoordinates(points, through, normal, a0, a1)));
    LightRectangle out;
    out.origin = (through + r.corner.x * a0) + r.corner.y * a1;
    out.u = r.side0.x * a0 + r.side0.y * a1;
    out.v = r.side1.x * a0 + r.side1.y * a1;
    return out;
}

// ---- transfer-function difference ---------------------------------------------------------------------------------------

namespace {

// Rule W -- how a transfer function is walked.  At step k its current node is node min(k, n - 1); the node coming up is node
// k + 1 as long as that is not the function's last node, and from then on a stand-in at position 1 with the last node's
// colour: the last node is met at 1, not where it lies.  The walk is spent after n steps.
class TfWalk {
public:
    explicit TfWalk(const std::vector<TfNode>& nodes) : nodes_(nodes), n_((long)nodes.size()) {}
    bool spent() const { return k_ >= n_; }
    void step() { ++k_; }
    const TfNode& current() const { return nodes_[(size_t)std::min(k_, n_ - 1)]; }
    TfNode coming() const {
        if (k_ + 2 < n_) return nodes_[(size_t)k_ + 1];
        TfNode atOne = nodes_.back();
        atOne.pos = 1.0;
        return atOne;
    }

private:
    const std::vector<TfNode>& nodes_;
    long n_, k_ = 0;
};

struct DifferencePoint {
    double pos = 0;
    float c[4] = { 0, 0, 0, 0 };
    float opacity() const { return c[3]; }
    bool differs(float eps) const { return std::fabs(c[0]) >= eps || std::fabs(c[1]) >= eps || std::fabs(c[2]) >= eps || std::fabs(c[3]) >= eps; }
};

struct ColorDistance {
    bool associated;
    // |q - p| per channel; with associated colours every channel (alpha too) is scaled by its own alpha first
    DifferencePoint at(double pos, const float p[4], const float q[4]) const {
        DifferencePoint d;
        d.pos = pos;
        const float wp = associated ? p[3] : 1.f, wq = associated ? q[3] : 1.f;
        for (int ch = 0; ch < 4; ++ch) d.c[ch] = std::fabs(q[ch] * wq - p[ch] * wp);
        return d;
    }
};

// the colour of the segment lo -> hi where it passes position x (lo.pos == hi.pos divides by zero, as it does in the reference)
void colourOnSegment(const TfNode& lo, const TfNode& hi, double x, float out[4]) {
    const double t = (x - lo.pos) / (hi.pos - lo.pos);
    for (int ch = 0; ch < 4; ++ch) out[ch] = (float)((double)lo.rgba[ch] * (1.0 - t) + (double)hi.rgba[ch] * t);
}

void append(TfBreakpoints& list, double pos, const float c[4]) {
    list.pos.push_back((float)pos);
    list.rgba.insert(list.rgba.end(), c, c + 4);
}

}  // namespace

// Rule D -- the list is |now - before| as a piecewise-linear function on [0, 1], reduced to what the importance kernel needs.
//   Opening point at 0: the distance of the two first nodes if the earlier of them lies beyond 0, one of them is not
//   transparent and the distance differs from zero; else zero.
//   The segment (from, to) under examination starts collapsed at the earlier first node with that distance; if both first nodes
//   are transparent and lie apart, `to` moves to the later one: its colour against the other function's colour there (that
//   function's first segment).
//   Then, while either walk (rule W) has steps left: a segment is LISTED if an end differs from zero and an end has opacity --
//   its `to` end, preceded by its `from` end when only the opening point is in the list yet (later gaps are bridged, not
//   re-opened); the next break is the earlier of the two coming nodes (both, if they coincide), valued as that node's colour
//   against the other function's colour on its current -> coming segment there.
//   Closing: the last `to` if it lies before 1 and has opacity; then a zero at 1 unless the list already reaches 1.
bool tfDifference(const std::vector<TfNode>& now, const std::vector<TfNode>& before, float epsilon, bool associatedColor, TfBreakpoints& out) {
    out.pos.clear();
    out.rgba.clear();
    const float none[4] = { 0.f, 0.f, 0.f, 0.f };
    if (now.empty() && before.empty()) {  // both positions are 0 in the reference's list for this case
        append(out, 0.0, none);
        append(out, 0.0, none);
        return true;
    }
    if (now.empty() || before.empty()) return false;
    const ColorDistance distance{ associatedColor };
    const TfNode &nowFirst = now.front(), &beforeFirst = before.front();
    float other[4];

    DifferencePoint from = distance.at(std::min(nowFirst.pos, beforeFirst.pos), nowFirst.rgba, beforeFirst.rgba);
    DifferencePoint to = from;
    if (nowFirst.pos != beforeFirst.pos && nowFirst.rgba[3] == 0.f && beforeFirst.rgba[3] == 0.f) {
        const bool nowLeads = nowFirst.pos < beforeFirst.pos;
        const std::vector<TfNode>& leading = nowLeads ? now : before;
        const TfNode& later = nowLeads ? beforeFirst : nowFirst;
        colourOnSegment(leading.front(), leading[std::min<size_t>(1, leading.size() - 1)], later.pos, other);
        to = distance.at(later.pos, later.rgba, other);
    }
    const bool opensLit = from.pos > 0. && (nowFirst.rgba[3] > 0.f || beforeFirst.rgba[3] > 0.f) && from.differs(epsilon);
    append(out, 0.0, opensLit ? from.c : none);

    TfWalk a(now), b(before);
    while (!a.spent() || !b.spent()) {
        if ((from.differs(epsilon) || to.differs(epsilon)) && (from.opacity() > 0.f || to.opacity() > 0.f)) {
            if (out.size() == 1) append(out, from.pos, from.c);
            append(out, to.pos, to.c);
        }
        const TfNode nextA = a.coming(), nextB = b.coming();
        from = to;
        if (nextA.pos < nextB.pos) {
            colourOnSegment(b.current(), nextB, nextA.pos, other);
            to = distance.at(nextA.pos, nextA.rgba, other);
            a.step();
        } else if (nextB.pos < nextA.pos) {
            colourOnSegment(a.current(), nextA, nextB.pos, other);
            to = distance.at(nextB.pos, nextB.rgba, other);
            b.step();
        } else {
            to = distance.at(nextA.rgba[3] < nextB.rgba[3] ? nextB.pos : nextA.pos, nextA.rgba, nextB.rgba);
            a.step();
            b.step();
        }
    }
    if (to.pos < 1. && to.opacity() > 0.f) append(out, to.pos, to.c);
    if (out.pos.back() < 1.f) append(out, 1.0, none);
    return true;
}

}  // namespace cpm_host

extern "C" {

void cpmh_fit_light_rectangle(const float* points_xyz, int n, const float through[3], const float unit_normal[3], float out9[9]) {
    std::vector<cpm_host::Pt3> pts((size_t)std::max(n, 0));
    for (size_t i = 0; i < pts.size(); ++i) pts[i] = { points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2] };
    const cpm_host::LightRectangle r = cpm_host::fitLightRectangle(pts, { through[0], through[1], through[2] }, { unit_normal[0], unit_normal[1], unit_normal[2] });
    const cpm_host::Pt3 parts[3] = { r.origin, r.u, r.v };
    for (int k = 0; k < 3; ++k) { out9[3 * k] = parts[k].x; out9[3 * k + 1] = parts[k].y; out9[3 * k + 2] = parts[k].z; }
}

int cpmh_hull_cycle(const float* points_xy, int n, float* cycle_xy) {
    std::vector<cpm_host::Pt2> pts((size_t)std::max(n, 0));
    for (size_t i = 0; i < pts.size(); ++i) pts[i] = { points_xy[2 * i], points_xy[2 * i + 1] };
    const std::vector<cpm_host::Pt2> cycle = cpm_host::hullCycle(pts);
    for (size_t i = 0; i < cycle.size(); ++i) { cycle_xy[2 * i] = cycle[i].x; cycle_xy[2 * i + 1] = cycle[i].y; }
    return (int)cycle.size();
}

void cpmh_smallest_rectangle(const float* cycle_xy, int n, float out6[6]) {
    std::vector<cpm_host::Pt2> cycle((size_t)std::max(n, 0));
    for (size_t i = 0; i < cycle.size(); ++i) cycle[i] = { cycle_xy[2 * i], cycle_xy[2 * i + 1] };
    const cpm_host::Rectangle2 r = cpm_host::smallestRectangle(cycle);
    const cpm_host::Pt2 parts[3] = { r.corner, r.side0, r.side1 };
    for (int k = 0; k < 3; ++k) { out6[2 * k] = parts[k].x; out6[2 * k + 1] = parts[k].y; }
}

int cpmh_tf_difference(const double* now_pos, const float* now_rgba, int n_now, const double* before_pos, const float* before_rgba, int n_before,
                       float epsilon, int associated_color, float* out_pos, float* out_rgba) {
    auto nodes = [](const double* pos, const float* rgba, int n) {
        std::vector<cpm_host::TfNode> v((size_t)std::max(n, 0));
        for (size_t i = 0; i < v.size(); ++i) {
            v[i].pos = pos[i];
            std::memcpy(v[i].rgba, rgba + 4 * i, sizeof v[i].rgba);
        }
        return v;
    };
    cpm_host::TfBreakpoints list;
    if (!cpm_host::tfDifference(nodes(now_pos, now_rgba, n_now), nodes(before_pos, before_rgba, n_before), epsilon, associated_color != 0, list)) return -1;
    std::memcpy(out_pos, list.pos.data(), list.pos.size() * sizeof(float));
    std::memcpy(out_rgba, list.rgba.data(), list.rgba.size() * sizeof(float));
    return (int)list.size();
}

}  // extern "C"
