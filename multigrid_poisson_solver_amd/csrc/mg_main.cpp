// MG_HIP -- command-line twin of the reference's MG_CPU / MG_GPU programs
// (README.md:130-139):   ./MG_HIP N_THREADS_OMP cycle_filename.txt
#include "mg_hip.h"

int main(int argc, char **argv) { return mg_cycle_main(argc, argv); }
