// Exchanges result tables with opencorr_amd/io.py through the reference's file formats (host only, no GPU).
//   io_driver <in_table2d.csv> <out_table2d.csv> <out_deformation2d.csv> <in_table3d.csv> <out_table3d.csv> <out_map.csv>
#include <iostream>

#include "opencorr_compat/opencorr.h"

using namespace opencorr;

int main(int argc, char** argv) {
    if (argc != 7) return 2;
    try {
        IO2D io2;
        io2.setDelimiter(",");
        io2.setPath(argv[1]);
        std::vector<POI2D> q2 = io2.loadTable2D();
        io2.setPath(argv[2]);
        io2.saveTable2D(q2);
        io2.setPath(argv[3]);
        io2.saveDeformationTable2D(q2);
        IO3D io3;
        io3.setDelimiter(",");
        io3.setPath(argv[4]);
        std::vector<POI3D> q3 = io3.loadTable3D();
        io3.setPath(argv[5]);
        io3.saveTable3D(q3);
        io2.setPath(argv[6]);
        io2.setWidth(12);
        io2.setHeight(9);
        io2.saveMap2D(q2, OutputVariable::e_yy);
        std::vector<Point2D> pts;
        for (const POI2D& p : q2) pts.push_back(p);
        io2.savePoint2D(pts, std::string(argv[6]) + ".points");
        if (io2.loadPoint2D(std::string(argv[6]) + ".points").size() != q2.size()) return 3;
    } catch (const std::string& msg) {
        std::cerr << msg << std::endl;
        return 1;
    }
    return 0;
}
