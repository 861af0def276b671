"""Python faces of the reference's compiled extensions that sit on the hot path (pcdet/ops/**), served by libtoda_hip.so."""
