"""Drop-in for the `simple_knn` package the reference's 2DGS adaptor imports at module import time
(/root/reference/lightning/renderer_2dgs.py:11 `from simple_knn._C import distCUDA2`; absent from the reference tree),
backed by the MI355X HIP library.  Implementation: generativedensification_amd/knn.py -> libgdr_hip.so."""
