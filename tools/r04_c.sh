#!/bin/bash
# round 4: phases of k_lattice_wave after the parallel character-type pass (KAMD_LATTICE_STOP), quick parity check
mkdir -p gpurun_out/r04_c; O=$PWD/gpurun_out/r04_c; ROOT=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lattices or tokens_bit or fuzzed" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python tools/lattice_phases.py c2-64k c4-cong > $O/lattice_phases.txt 2>&1; cat $O/lattice_phases.txt
