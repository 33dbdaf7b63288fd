// scene_dev.h — the flattened scene as it lives in HBM (structure-of-arrays primitive buffers,
// packed material table, light list).  Passed to kernels by value (kernarg).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct MatDev {   // packed material table row (32 B); kinds as in cftrt.h
  int kind;
  float p0, p1;
  float r, g, b;
  float pad0, pad1;
};

struct LightDev {  // device light list row
  int kind;
  int sph_begin, sph_count;
  int tri_begin, tri_count;
  float vx, vy, vz;   // pos (point) / dir (parallel)
  float r, g, b;
};

// Primitive ids.  A "global prim id" g names one primitive of the input arrays:
//   g in [0, n_sph)            -> sphere g
//   g in [n_sph, n_sph+n_tri)  -> triangle g - n_sph
// prim_obj[g] is the owning object's index in world.objects (the parity "object id").
struct SceneDev {
  // ---- f32 cull lists (hot loops) ----
  const float4* sphF;     // static, non-precise spheres: (-2c'x, -2c'y, -2c'z, 4(|c'|^2 - r^2)), c' = c - bc
  const float* sphR2;     // 16 * their r^2 (f32-rounded r^2 scaled exactly), padded like sphF (fetched 4 at a time beside each sphF batch)
  const int* sphF_gid;    // -> global prim id
  int n_sphF;
  const float4* msphF;    // moving spheres, 2 float4 each: (cx, cy, cz, r^2) at time 0, (mvx, mvy, mvz, object id bits)
  const float4* msphB;    // their swept bounds in sphF's format (-2 b', 4 (|b'|^2 - rb^2)), padded like sphF
  const int* msph_gid;
  int n_msph;
  const int* prec_gid;    // "precise" spheres (|r| >= precise_radius): exact f64 test for every ray
  int n_prec;
  const float4* triF;     // static triangles, 3 float4 each: (p0, -), (e1, -), (e2, -)
  const int* triF_gid;
  int n_triF;
  // the same triangles as the ITEMS the scans test: a single triangle, or the two triangles of a Square::new as one
  // parallelogram (kernels.h, tri_item_bounds).  3 float4 each: (p0, meta bits), (e1 | ea), (e2 | eb); meta = (index in
  // triF of the item's first triangle << 1) | quad; a quad's second triangle is the next entry of triF.
  const float4* triItem;
  int n_item, nb_item;    // the first nb_item items are too big for the grid (scanned linearly by every ray)
  const float4* mtriF;    // moving triangles, 4 float4 each: p0, e1, e2, (mv, object id bits)
  const int* mtri_gid;
  int n_mtri;
  // ---- uniform grid over the lists above (an extension; 0 cells = none).  Cell (x, y, z) -> gridCell[(z*gny + y)*gnx + x]
  const uint2* gridCell;  // x = (first sphere entry << 8) | count; y = (first triangle entry << 12) | (count << 4)
  const float4* gridSph;  // per entry: the sphF / msphB entry
  const float2* gridSphX; // per entry: (16 r^2, list index bits), read when the cull passes
  const uint32_t* gridTri;   // index into triItem
  // The sphere part of the grid once more as ONE image, for the walk instances that keep it in LDS (a scene whose grid
  // fits, CFT_LDS_GRID_BYTES = 50 KB: S500's is 35 KB): [sphere entries, 16 B each][cell records, 8 B each][(16 r^2, list
  // index), 8 B each].  Null: too big, or the cells list triangles.
  const uint4* gridImg;
  uint32_t gridImgBytes, gridImgEnt;   // size (a multiple of 16) and the number of sphere entries
  int gnx, gny, gnz;
  int nb_sph, nb_msph, nb_tri;   // the first nb_* entries of sphF / msphF+msphB / triF are too big for the grid: scanned linearly
  int grid_mode;          // 0: lists only; 1: waves too wide for the cone cull walk the grid; 2: every wave does
  float g0x, g0y, g0z, g1x, g1y, g1z;   // the (inflated) box, relative to the scene centre
  float gcsx, gcsy, gcsz, gicx, gicy, gicz;   // cell size and its inverse
  float gfar4;            // 4 * the largest |coordinate| of a ray origin the walk accepts
  // ---- f64 master copies (exact evaluation) ----
  const double* sphD;     // n_sph x 4: cx, cy, cz, r
  const double* sphMvD;   // n_sph x 3 (all zero for static spheres)
  int n_sph;
  const double* triD;     // n_tri x 9: p0, p1, p2
  const double* triMvD;   // n_tri x 3
  int n_tri;
  // Square::from_points (object.rs:169-174): its hit is tri0.or(tri1) (object.rs:192) - the second triangle counts only
  // where the first is not hit, whatever the distances.  tri_first[i] = the triangle (index into triD, same object)
  // that shadows triangle i this way, -1 for every other triangle; likewise ltri_first for the light shapes' triangles.
  const int* tri_first;
  const int* ltri_first;
  int any_pairs;          // some entry of tri_first / ltri_first is >= 0
  // ---- the listed primitives again, in LIST order (index = what the f32 scan hands over), so that the exact evaluation of
  // a candidate starts with one round of loads instead of list index -> prim id -> data -> material id:
  const double* lsD;      // static list spheres, 4 each: cx, cy, cz, r
  const double* lmD;      // moving list spheres, 8 each: cx, cy, cz, r, mvx, mvy, mvz, -
  const int4* listRec;    // every list entry [sphF | msph | triF | mtri]: (global prim id, object id, material id, -)
  // the first precise sphere, in the kernel arguments themselves (the ground of scene.rs wins most rays)
  double prec_c0x, prec_c0y, prec_c0z, prec_r0;
  int prec_g0, prec_mat0;   // prec_g0 = -1: none, or it moves
  // ---- per-prim tables ----
  const int* prim_obj;
  const int* prim_mat;
  const MatDev* mats;
  int n_mat;
  // ---- lights ----
  const LightDev* lights;
  int n_lights;
  const double* lsphD;    // light-shape spheres, 4 each
  const double* ltriD;    // light-shape triangles, 9 each
  int wave_cull;          // 1: the level kernels may cull spheres per wave (cftrt_ctx_set_wave_cull)
  // ---- per-tile lists for the primary rays (set per render call; null: none) ----
  const float4* boundF;   // every entry of [sphF | msph | triF]: a bounding sphere (centre relative to the scene centre, radius)
  const uint32_t* tile_off;   // n_tiles + 1 offsets into tile_idx (tile = 32x8 pixels, row-major over the tiles)
  const uint32_t* tile_idx;   // list indices (into [sphF | msph | triF]) of the primitives a ray of the tile may reach
  int any_motion;
  int prec_motion;        // some precise sphere moves
  int n_phong_mats;
  // ---- f32 error model of the cull, set by cftrt_scene_commit ----
  float K1, K2;            // |disc - true disc| <= K1*|q| + K2
  float K3, K4;            // |q - true q|       <= K3*|q| + K4
  float Kc, rmax2;         // cheap form: |disc' - true| <= Kc*((|o'| + brad)^2 + rmax^2)
  float bcx, bcy, bcz, brad;   // scene centre bc (f32 lists are stored relative to it) and the largest |c'| (+ motion)
  float tKA;               // |a - true a| (triangle determinant)
  float tKU1, tKU2;        // |u,v - true| <= |f| * (tKU1*|s|_1 + tKU2)
  float tKT1, tKT2, tKT3;  // |t - true t| <= |f| * (tKT1*|s|_1 + tKT2 + tKT3*|t|)
};

// Ray queue slot: 40 B in three coalesced streams.
struct QueueDev {
  float4* a;   // ox, oy, oz, dx
  float4* b;   // dy, dz, thr_r, thr_g
  float2* c;   // thr_b, path id (uint bits)
};
// Hit record: 40 B in three streams.
struct HitDev {
  float4* h0;  // t (f32-rounded), prim (int bits; -1 = miss), px, py
  float4* h1;  // pz, nx, ny, nz   (geometric unit normal before HitInfo::new's flip)
  double* t;   // the exact f64 distance (+inf on a miss): LightShape::looked compares distances in f64 (light.rs:236)
};
// The records of the lights' `looked` scans (split pipeline).  One record set serves every light of a static scene (the
// scans are identical) and the one looked-capable light of a moving scene; a MOVING scene with several looked-capable
// lights (SkyLight, LightShape) holds one set per such light, in the lights' order: the reference scans the world once
// per light that looks, each scan with its own motion times (object.rs:333-338, light.rs:184-190,233-242).
#define CFT_MAX_LOOK 8
struct LookRecs {
  HitDev r[CFT_MAX_LOOK];
  int n;
};
