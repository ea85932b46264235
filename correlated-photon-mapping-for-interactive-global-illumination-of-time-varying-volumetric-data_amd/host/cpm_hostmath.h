// cpm_hostmath.h -- the host arithmetic of the path that decides device inputs, free of Inviwo types and of HIP:
//
//   * the light's rectangle: where a directional light's sample lattice lies.  The reference fits it on the CPU
//     (lightcl/orientedboundingbox2d.cpp:80-100 and what it calls); the emitted photon set -- and with it every RNG
//     stream's sample -- follows from its origin and edges, so the RESULT has to be the reference's, ties included.
//   * the break points of |TF_new - TF_old|: what the importance kernel classifies bricks with after a
//     transfer-function edit (importancesamplingcl/processors/minmaxuniformgrid3dimportanceclprocessor.cpp:364-501).
//
// Both are written from their rules (stated at the definitions), not from the reference's text; tests/test_host_logic.py
// holds them to the statement-by-statement restatement kept with the checker (oracle/, its host part) on random and on tied inputs.
// The processors (cpm_processors.cpp) and the Python driver (pipeline.py, through the C entry points below) use these.
#pragma once
#include <cstddef>
#include <vector>

namespace cpm_host {

struct Pt2 { float x = 0, y = 0; };
struct Pt3 { float x = 0, y = 0, z = 0; };

// ---- the light's rectangle ------------------------------------------------------------------------------------------

// Coordinates of `points` in the plane through `through` with unit normal `normal`, along the in-plane unit axes a0, a1.
std::vector<Pt2> planeCoordinates(const std::vector<Pt3>& points, Pt3 through, Pt3 normal, Pt3 a0, Pt3 a1);

// The vertex cycle whose edges the rectangle search tries, in the order it tries them (rule H at the definition).
std::vector<Pt2> hullCycle(std::vector<Pt2> points);

struct Rectangle2 { Pt2 corner, side0, side1; };
// Smallest-area rectangle with a side along an edge of `cycle` (closing edge first, an earlier edge wins a tie).
Rectangle2 smallestRectangle(const std::vector<Pt2>& cycle);

struct LightRectangle { Pt3 origin, u, v; };
// normal: unit length.  The in-plane axes are fixed by the rule of orientedboundingbox2d.cpp:82-87.
LightRectangle fitLightRectangle(const std::vector<Pt3>& points, Pt3 through, Pt3 normal);

// ---- transfer-function difference -------------------------------------------------------------------------------------

struct TfNode {
    double pos = 0;
    float rgba[4] = { 0, 0, 0, 0 };
};
struct TfBreakpoints {
    std::vector<float> pos;
    std::vector<float> rgba;  // 4 per point
    size_t size() const { return pos.size(); }
};
// now / before: sorted by position.  false when exactly one of them is empty (there is no difference function then; the
// caller classifies with the transfer function itself).
bool tfDifference(const std::vector<TfNode>& now, const std::vector<TfNode>& before, float epsilon, bool associatedColor, TfBreakpoints& out);

}  // namespace cpm_host

// Plain-C doors for callers outside C++ (pipeline.py): same results, no GPU touched.
extern "C" {
// points_xyz: n x 3; out9 = origin.xyz, u.xyz, v.xyz
void cpmh_fit_light_rectangle(const float* points_xyz, int n, const float through[3], const float unit_normal[3], float out9[9]);
// cycle_xy: room for n + 2 points; returns the number of vertices
int cpmh_hull_cycle(const float* points_xy, int n, float* cycle_xy);
// out6 = corner.xy, side0.xy, side1.xy
void cpmh_smallest_rectangle(const float* cycle_xy, int n, float out6[6]);
// outputs: room for n_now + n_before + 2 points; returns the count, -1 when exactly one function is empty
int cpmh_tf_difference(const double* now_pos, const float* now_rgba, int n_now, const double* before_pos, const float* before_rgba, int n_before,
                       float epsilon, int associated_color, float* out_pos, float* out_rgba);
}
