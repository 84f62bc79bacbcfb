"""raytracing_amd — MI355X (gfx950) path-tracing core behind the reference's GL seam.

Host-side mirror of the reference's operator interface for the hot path:
  SSBO.Bind                 <- SSBO::Bind                (SSBO.cpp:5-17)
  ComputeShader.set*        <- ComputeShader::set*       (computeShader.cpp:53-75)
  ComputeShader.dispatch    <- the glDispatchCompute tile loop (main.cpp:126-141)
  ComputeShader.get_image   <- glGetTexImage             (main.cpp:157)
all of which call the C ABI in include/rt_core.h (librt_core.so, hand-written HIP).
"""
from .core import ComputeShader, SSBO, RtError  # noqa: F401
