"""Removed in round 4 (ABI 5).  The driver-side stand-in `HipTrainer` is test infrastructure and lives outside the product
package now; the product entry point is `isdf_amd.hot_path.graft(trainer)` on the reference's own `Trainer`
(INTEGRATION.md section A).  This stub exists so that an old `import isdf_amd.trainer` fails with directions instead of a bare
ModuleNotFoundError."""
raise ImportError(
    "isdf_amd.trainer was removed (round 4): graft the reference's Trainer with isdf_amd.hot_path.graft(trainer) "
    "(INTEGRATION.md section A); checkpoints: isdf_amd.hot_path.hip_state_dict / load_hip_state_dict; the driver-side stand-in used "
    "by the benchmarks and tests is bench_support/standin_trainer.py (outside the product package)")
