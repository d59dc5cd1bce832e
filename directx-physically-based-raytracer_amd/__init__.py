"""MI355X-native wavefront path tracer: drop-in for the reference's GBufferGeneration + Raytracing path.

The directory name is not a Python identifier; load it with __graft_entry__.load_package(), which
registers it as ``dxpbrt_amd``. Submodules:
  layouts  numpy mirrors of the reference's data layouts
  scenes   procedural BASELINE.json scenes in those layouts
  ptamd    ctypes binding of the C-ABI (include/ptamd.h) + the Python mirror of the reference operators
"""
