// raytracer_3d_gpu — drop-in for the reference executable built from src/raytracer_3d.cu:
// `raytracer_3d_gpu <n>` shoots 2^n photons through the built-in 128^3 block-cloud scene
// (raytracer_3d.cu:11-65) and writes the seven tally files into the working directory.
#include "driver_common.h"

int main(int argc, char* argv[])
{
    drv::Config c;
    c.itot = c.jtot = c.ktot = 128;                 // raytracer_3d.cu:15-17
    c.dx = c.dy = 50.f; c.dz = 25.f;                // raytracer_3d.cu:11-13
    drv::parse_args(argc, argv, c);
    if (c.exponent < 15)
    {
        // raytracer_3d.cu:217-222 (its fixed 64 x 512 launch needed one photon per thread)
        std::cerr << "Sorry, the number of photons must be larger than " << 64 * 512 << " (n >= 15) to guarantee one photon per thread" << std::endl;
        return 1;
    }
    try
    {
        const float k_ext_gas = 1.e-4f, ssa_gas = 0.5f, asy_gas = 0.f;         // raytracer_3d.cu:33-35
        const float k_ext_cloud = 5.e-3f, ssa_cloud = 0.9f, asy_cloud = 0.85f; // raytracer_3d.cu:37-39
        const size_t n = (size_t)c.itot * c.jtot * c.ktot;
        std::vector<float> kg(n, k_ext_gas), kc(n, 0.f), ssa(n, ssa_gas), asy(n, asy_gas);
        for (int k = 0; k < c.ktot; ++k)
            for (int j = 0; j < c.jtot; ++j)
                for (int i = 0; i < c.itot; ++i)
                    if ((i + 0.5) * c.dx > 4000. && (i + 0.5) * c.dx < 5000. && (j + 0.5) * c.dy > 2700. && (j + 0.5) * c.dy < 3700.
                        && (k + 0.5) * c.dz > 1000. && (k + 0.5) * c.dz < 1500.)
                    {
                        const size_t ijk = i + (size_t)j * c.itot + (size_t)k * c.itot * c.jtot;
                        kc[ijk] = k_ext_cloud;
                        ssa[ijk] = (ssa_gas * k_ext_gas + ssa_cloud * k_ext_cloud) / (k_ext_gas + k_ext_cloud);
                        asy[ijk] = (asy_gas * ssa_gas * k_ext_gas + asy_cloud * ssa_cloud * k_ext_cloud)
                                 / (ssa_gas * k_ext_gas + ssa_cloud * k_ext_cloud);
                    }
        drv::run(c, kg, kc, ssa, asy);
    }
    catch (const std::exception& e)
    {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
