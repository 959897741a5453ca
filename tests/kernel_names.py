"""Names of the kernels the library picks by default, in one place for the tests that assert the selection."""
LDS_STEPPER = 'qgs_spec_rkldsa8'        # LDS-resident stepper of rank-3 tensors: hand-scheduled stage body (codegen_lds_asm.cpp)
LDS_STEPPER_RANK5 = 'qgs_spec_rklds16'   # ... of rank-5 tensors (derived monomials): compiler-scheduled (codegen_lds.cpp)
