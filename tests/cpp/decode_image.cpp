// Decodes an image file with the drop-in Image2D(path) / a volume with Image3D(path) and dumps the floats row-major
// (tests/test_reference_examples_compile.py compares them with numpy).  usage: decode_image 2d|3d <in> <out.bin>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "opencorr_compat/opencorr.h"

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    try {
        std::vector<float> out;
        int dims[3] = {0, 0, 0};
        if (!strcmp(argv[1], "2d")) {
            opencorr::Image2D img{std::string(argv[2])};
            dims[0] = img.width; dims[1] = img.height; dims[2] = 1;
            for (int r = 0; r < img.height; r++)
                for (int c = 0; c < img.width; c++) out.push_back(img.eg_mat(r, c));
        } else {
            opencorr::Image3D vol{std::string(argv[2])};
            dims[0] = vol.dim_x; dims[1] = vol.dim_y; dims[2] = vol.dim_z;
            out.assign(&vol.vol_mat[0][0][0], &vol.vol_mat[0][0][0] + vol.size);
        }
        FILE* f = fopen(argv[3], "wb");
        if (!f) return 3;
        fwrite(dims, sizeof(int), 3, f);
        fwrite(out.data(), sizeof(float), out.size(), f);
        fclose(f);
    } catch (std::string& e) {
        fprintf(stderr, "%s\n", e.c_str());
        return 4;
    }
    return 0;
}
