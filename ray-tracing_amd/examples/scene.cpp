// scene.cpp — examples/scene.rs of the reference, restated over the C++ host mirror: the random
// "one weekend" scene (488 objects, SkyLight, thin-lens camera), rendered by ONE World::render call
// instead of the hand-rolled thread-pool loop (scene.rs:86-102).
//   usage: scene [width height spp depth out.ppm [raw_accum.f32]]      defaults: 800 500 50 10 test.ppm
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "../cpp/cft_ray_tracer.hpp"
using namespace cft;

int main(int argc, char** argv) {
  const uint64_t WIDTH = argc > 1 ? strtoull(argv[1], 0, 10) : 800, HEIGHT = argc > 2 ? strtoull(argv[2], 0, 10) : 500;
  const uint64_t SAMPLE_RATE = argc > 3 ? strtoull(argv[3], 0, 10) : 50, TRACE_DEPTH = argc > 4 ? strtoull(argv[4], 0, 10) : 10;
  const char* out = argc > 5 ? argv[5] : "test.ppm";
  World world = World::empty();
  LambertianModel d(1.0);
  Dielectric t(1.5);
  Metal m(0.3, 1.0);
  world.add_obj(Object(Sphere({0., 0., -1000.}, 1000.), d.with_color({0.5, 0.5, 0.5})));
  Rd rd(0x5EED0001ull);   // the reference's thread_rng() is unseeded (scene.rs:34)
  for (int a = -11; a < 11; ++a) {
    for (int b = -11; b < 11; ++b) {
      double cx = a + 0.9 * rd(), cy = b + 0.9 * rd();
      Vec3 center(cx, cy, 0.2);
      double choose_material = rd();
      Object obj = [&]() -> Object {
        if (choose_material < 0.8) {
          double r = rd(), g = rd(), bl = rd();
          return Object(Sphere(center, 0.2), d.with_color({r * r, g * g, bl * bl}));
        } else if (choose_material < 0.95) {
          double r = rd(), g = rd(), bl = rd(), fz = rd();
          return Object(Sphere(center, 0.2), m.with_color({(1. + r) / 2., (1. + g) / 2., (1. + bl) / 2.}).with_fuzz(fz / 2.));
        }
        return Object(Sphere(center, 0.2), t);
      }();
      double choose_move = rd();
      world.add_obj(choose_move < 0.4 ? obj.moved({0., 0., 0.1}) : obj);
    }
  }
  world.add_obj(Object(Sphere({0., 0., 1.}, 1.), t));
  world.add_obj(Object(Sphere({-4., 0., 1.}, 1.), d.with_color({0.4, 0.2, 0.1})));
  world.add_obj(Object(Sphere({4., 0., 1.}, 1.), m.with_color({0.7, 0.6, 0.5}).with_fuzz(0.)));
  world.add_light(SkyLight());

  Camera camera = Camera({13., -3., 2.}, {0., 0., 0.})
                      .with_focus_dist(10.).with_aperture(0.1).with_fov(20.)
                      .with_aspect((double)WIDTH / (double)HEIGHT).with_sample_rate(SAMPLE_RATE);
  Device dev(0);
  auto start = std::chrono::steady_clock::now();
  cftrt_stats st;
  std::vector<Color> raw = world.render(camera, WIDTH, HEIGHT, TRACE_DEPTH, dev, 0x5EED0002ull, &st);
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
  printf("total: %.3f seconds, %.3f ns/pixel   (%llu rays, %.1f Mrays/s on the device)\n", secs,
         secs * 1e9 / (double)(WIDTH * HEIGHT * SAMPLE_RATE), (unsigned long long)st.rays, st.rays / st.ms_total / 1e3);
  std::ofstream f(out, std::ios::binary);
  f << "P6\n" << WIDTH << " " << HEIGHT << "\n255\n";
  for (const Color& p : raw) {
    uint8_t rgb[3];
    vec3_to_rgb(p / (double)SAMPLE_RATE, rgb);
    f.write((const char*)rgb, 3);
  }
  if (argc > 6) {
    std::ofstream r(argv[6], std::ios::binary);
    for (const Color& p : raw) { float v[3] = {(float)p.x, (float)p.y, (float)p.z}; r.write((const char*)v, 12); }
  }
  return 0;
}
