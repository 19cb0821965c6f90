// raytracer_3d_io_gpu — drop-in for the reference executable built from src/raytracer_3d_io.cu:
// reads k_ext_gas.bin, k_ext_cloud.bin, ssa.bin, asy.bin (headerless float32 [k][j][i], the
// make_binary.py format) from the working directory (or --indir=), default grid 240 x 240 x 284
// at 100/100/30 m (raytracer_3d_io.cu:11-17), and writes the seven tally files. A sidecar grid.hdr in
// the input directory (dimensions, spacings, dtype; see driver_common.h) replaces those compiled-in
// defaults; --itot= ... options override both.
// `--zenith=a,b,c` sweeps solar zenith angles with the grids resident on the GPU, one output
// directory per angle (config "sweep of solar zenith angles" of BASELINE.json).
#include "driver_common.h"

int main(int argc, char* argv[])
{
    drv::Config c;
    c.itot = 240; c.jtot = 240; c.ktot = 284;       // raytracer_3d_io.cu:15-17
    c.dx = c.dy = 100.f; c.dz = 30.f;               // raytracer_3d_io.cu:11-13
    drv::read_sidecar(drv::find_indir(argc, argv), c);
    drv::parse_args(argc, argv, c);
    if (c.exponent < 15)
    {
        std::cerr << "Sorry, the number of photons must be larger than " << 64 * 512 << " (n >= 15) to guarantee one photon per thread" << std::endl;
        return 1;
    }
    try
    {
        const size_t n = (size_t)c.itot * c.jtot * c.ktot;
        std::vector<float> kg(n), kc(n), ssa(n), asy(n);
        const auto load = c.input_f64 ? drv::load_binary_f64 : drv::load_binary;
        load(c.indir, "k_ext_gas", kg.data(), n);       // raytracer_3d_io.cu:54-57
        load(c.indir, "k_ext_cloud", kc.data(), n);
        load(c.indir, "ssa", ssa.data(), n);
        load(c.indir, "asy", asy.data(), n);
        drv::run(c, kg, kc, ssa, asy);
    }
    catch (const std::exception& e)
    {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
