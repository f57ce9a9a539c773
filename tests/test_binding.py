"""The reference-side binding (bindings/smarties/RACER_HIP.h: `class RACER_HIP : public smarties::Learner`), compiled inside
the reference tree and linked with the reference's own objects by `make -C oracle binding` (build container only; the binary
travels to the GPU box under oracle/_ref/).  On the GPU it is driven the way Core/Worker.cpp drives a learner -- agents through
Learner::select, the TaskQueue through setupTasks -- next to the reference's own V-RACER fed the same observations."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "binding_check")


def test_binding_source_is_present_and_cites_the_interface():
    src = open(os.path.join(ROOT, "bindings", "smarties", "RACER_HIP.h")).read()
    for needle in ("class RACER_HIP : public Learner", "void selectAction(const MiniBatch& MB, Agent& agent) override",
                   "void processTerminal(const MiniBatch& MB, Agent& agent) override", "void setupTasks(TaskQueue& tasks) override",
                   "hl_append_episode", "hl_step", "hl_forward"):
        assert needle in src, needle


@pytest.mark.gpu
def test_compiled_binding_trains_inside_the_reference_process(tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/binding_check not built (needs /root/reference: make -C oracle binding)")
    out = subprocess.run([EXE, "200"], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["nparams_ref"] == r["nparams_hip"]
    assert r["stored_ref"] == r["stored_hip_host"] == r["stored_hip_device"]          # every episode of the plug-in path reached the device
    assert r["steps_ref"] == r["steps_hip"] == 200
    # same observations, same initial weights, same generator for the actions: the two learners see the same replay; their
    # minibatches differ only through the episode order of the sampling table, so the ReF-ER state and the weights stay close
    assert abs(r["beta_ref"] - r["beta_hip"]) < 1e-3 * r["beta_ref"]
    assert abs(r["wnorm_ref"] - r["wnorm_hip"]) < 2e-3 * r["wnorm_ref"]
    assert "hip stats line:" in out.stdout and os.path.exists(tmp_path / "hip_00_stats.txt") is False   # (no print step reached in 200)


@pytest.mark.gpu
def test_compiled_binding_resumes_like_the_reference(tmp_path):
    """Core/Worker.cpp:291-295 on a restarted process: restart(), then setupTasks(), whose first task is initializeLearner() again
    -- skipped for a restarted learner (Learner.cpp:51-54): the ReF-ER state read from the checkpoint must survive it.  The
    binding resumes from its own files and from the files the reference's learner wrote, next to the reference resuming from
    those."""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/binding_check not built (needs /root/reference: make -C oracle binding)")
    out = subprocess.run([EXE, "150", "restart", "100"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{\"restart\"")][-1])
    # the reference writes nGradSteps + 1 into its status file (MemoryBuffer.cpp:183): both resume one step "later" than they saved
    assert r["grad0_ref"] == r["grad0_hip"] == r["grad0_x"] == 151
    assert r["beta_restarted"] == pytest.approx(r["beta_saved"], rel=1e-6)       # (the status file holds six digits, MemoryBuffer.cpp:313)
    assert r["beta_after_init_task"] == r["beta_restarted"]                 # the init task left the restored state alone
    assert r["beta_x_restarted"] == pytest.approx(r["beta_ref_saved"], rel=1e-6) and r["beta_x_after_init_task"] == r["beta_x_restarted"]
    assert r["stored_restarted"] == r["stored_saved"] and r["stored_x"] == r["stored_ref"]
    assert r["steps_ref"] == r["steps_hip"] == r["steps_x"] == 251
    assert abs(r["beta_ref"] - r["beta_x"]) < 2e-3 * r["beta_ref"] and abs(r["beta_ref"] - r["beta_hip"]) < 2e-3 * r["beta_ref"]
    assert abs(r["wnorm_ref"] - r["wnorm_x"]) < 2e-3 * r["wnorm_ref"] and abs(r["wnorm_ref"] - r["wnorm_hip"]) < 2e-3 * r["wnorm_ref"]
