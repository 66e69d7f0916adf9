// CPU: the C++ host's load_from_file / save_to_file (nz_io.cpp; reference main.cpp:157-233) against the Python restatement:
//     test_nz_io <in.nz> <out.nz>     loads <in.nz> (written by unet-studio_amd/nz.py), prints the fields, saves to <out.nz>
#include "unet.hpp"
#include <iostream>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    UNet3d model;
    if (!load_from_file(model, argv[1])) { std::cout << "LOAD FAILED" << std::endl; return 1; }
    std::cout << "in " << model->in_count << " out " << model->out_count << " params " << model->parameters().size() << std::endl;
    std::cout << model->get_info();
    std::cout << "errors " << model->training_errors.size() << " " << model->testing_errors.size() << std::endl;
    if (!save_to_file(model, argv[2])) { std::cout << "SAVE FAILED" << std::endl; return 1; }
    std::cout << "OK" << std::endl;
    return 0;
}
