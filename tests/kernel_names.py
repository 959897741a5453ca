"""Names of the kernels the library picks by default, in one place for the tests that assert the selection."""
LDS_STEPPER = 'qgs_spec_rkldsa8'        # LDS-resident stepper of rank-3 tensors: hand-scheduled stage body (codegen_lds_asm.cpp)
LDS_STEPPER_RANK5 = 'qgs_spec_rklds16'   # ... of rank-5 tensors (derived monomials): compiler-scheduled (codegen_lds.cpp)
TGL_PAIR = 'qgs_spec_tglp_s%d'             # tangent kernel on the paired stage record (compiler-scheduled, codegen_tangent.cpp)
TGL_PAIR_ASM = 'qgs_spec_tglpa_s%d'        # ... its hand-scheduled twin (QGS_HIP_TGL_ASM=1; rank-3 tensors, 2 - 4 stages, ndim <= 37)
LDS_TANGENT, LDS_ADJOINT = 'qgs_spec_tglldsa8', 'qgs_spec_adjldsa8'            # LDS-resident tangent / adjoint kernels of rank-3 tensors: hand-scheduled
LDS_TANGENT_RANK5, LDS_ADJOINT_RANK5 = 'qgs_spec_tgllds16', 'qgs_spec_adjlds16'  # ... of rank-5 tensors: compiler-scheduled (codegen_lds.cpp)
LDS_TANGENT_ASM, LDS_ADJOINT_ASM = 'qgs_spec_tglldsa8', 'qgs_spec_adjldsa8'    # hand-scheduled twins (the default; codegen_lds_asm.cpp)
LDS_TANGENT_CC, LDS_ADJOINT_CC = 'qgs_spec_tgllds16', 'qgs_spec_adjlds16'      # compiler-scheduled (QGS_HIP_LDS_TGL_ASM=0)
