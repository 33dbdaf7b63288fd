// cornell.cpp — examples/cornell.rs of the reference: the camera sits inside a Lambertian cube lit by a
// square LightShape under the ceiling; default camera (fov 45 *radians*, aspect 1 on a 400x300 image).
//   usage: cornell [width height spp depth out.ppm [raw_accum.f32]]      defaults: 400 300 5 10 test.ppm
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "../cpp/cft_ray_tracer.hpp"
using namespace cft;

int main(int argc, char** argv) {
  const uint64_t WIDTH = argc > 1 ? strtoull(argv[1], 0, 10) : 400, HEIGHT = argc > 2 ? strtoull(argv[2], 0, 10) : 300;
  const uint64_t SAMPLE_RATE = argc > 3 ? strtoull(argv[3], 0, 10) : 5, DEPTH = argc > 4 ? strtoull(argv[4], 0, 10) : 10;
  const char* out = argc > 5 ? argv[5] : "test.ppm";
  World world = World::empty();
  LambertianModel d(0.8);
  world.add_obj(Object(Cube({0., 0., 0.}, {1., 0., 0.}, {0., 1., 0.}, 2.), d));
  world.add_light(LightShape(Square({0., 0., 0.99}, {1., 0., 0.}, {0., -1., 0.}, 0.9)));
  Camera camera = Camera({0.8, 0.0, 0.0}, {0., 0., 0.0}).with_sample_rate(SAMPLE_RATE);
  Device dev(0);
  std::vector<Color> raw = world.render(camera, WIDTH, HEIGHT, DEPTH, dev);
  std::ofstream f(out, std::ios::binary);
  f << "P6\n" << WIDTH << " " << HEIGHT << "\n255\n";
  for (const Color& p : raw) {
    uint8_t rgb[3];
    vec3_to_rgb(p / (double)SAMPLE_RATE, rgb);   // every pixel's count is SAMPLE_RATE (cornell.rs:38,44)
    f.write((const char*)rgb, 3);
  }
  if (argc > 6) {
    std::ofstream r(argv[6], std::ios::binary);
    for (const Color& p : raw) { float v[3] = {(float)p.x, (float)p.y, (float)p.z}; r.write((const char*)v, 12); }
  }
  return 0;
}
