"""Drop-in for the one nvdiffrast entry point the reference uses (`import nvdiffrast.torch as dr`, models/modules.py:10;
`dr.texture(..., boundary_mode='cube')`, :202)."""
