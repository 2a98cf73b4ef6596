"""climsim_amd: MI355X-native training engine for ClimSim's baseline column emulators.

  climsim_amd.data_utils   - climsim_utils.data_utils-compatible loader/normalisation/metrics API
  climsim_amd.mlp          - MLPEmulator (Keras-like fit/predict over the HIP engine)
  climsim_amd.assets       - dependency-free readers for the grid / norm assets
  climsim_amd.build        - hipcc build of libclimsim_hip.so (gfx950)
"""
__version__ = "0.1.0"
