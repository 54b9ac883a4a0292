"""CPU oracle for the BoxInst mask-loss path -- TEST INFRASTRUCTURE ONLY.

``oracle.c_oracle``     : ctypes binding of the plain-C restatement (``boxinst_oracle.c``).
``oracle.torch_oracle`` : the same path in torch CPU ops (the reference's own CPU-runnable form).
``oracle.reference_extract`` : pulls the reference's torch-only functions out of
                          ``/root/reference`` by AST (build container only; used to make/verify
                          the golden fixtures in ``tests/golden``).

Nothing in ``boxinstseg_amd`` imports this package.
"""
