"""CPU tests of bench.py's process model (no GPU touched: --rehearse-cpu).

`python bench.py --gpus N` with no launcher must start N fresh rank processes itself, relay
rank 0's single JSON line and return the ranks' worst exit code; a collective that stalls must be
visible in the line ("stalled": true) AND in the exit code -- never rc 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(*flags, timeout=240, **env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, BENCH, *flags], capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return r.returncode, lines, r.stderr


def test_self_launch_two_ranks_prints_one_line():
    rc, lines, err = run("--gpus", "2", "--rehearse-cpu", "--backend", "gloo", "--steps", "5", "--warmup", "1")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 1
    assert line["value"] is None and "rehearsal" in line  # a rehearsal can never be mistaken for a measurement
    assert line["config"]["control_backend"] == "gloo" and line["rccl_ok"] is None
    # the N>1 line carries config 4's side measurements
    assert line["allgather_packed"]["own_slot_ok"] is True
    assert line["allgather_packed"]["all_slots_ok"] is True  # EVERY rank's slot of the gathered buffer against what it must hold
    assert "encode_allgather_end_to_end" in line
    assert "stalled" not in line and "collective_error" not in line
    # north_star's own split: a ragged batch by whole sequences, gather of unequal word counts (here: the exchange alone, slots hold their word indices)
    rb = line["ragged_batch_sharded"]
    assert rb["all_slots_ok"] is True and sum(rb["words_per_rank"]) == rb["total_words"] and len(set(rb["words_per_rank"])) == 2


def test_a_wrong_word_in_a_peer_slot_fails_the_run():
    """Config 4's check is literal: the gathered buffer is compared slot by slot with what every rank's shard must pack to (here, without a
    GPU, arange + rank), so a gather that delivered garbage into a PEER's slot is caught -- the line says so and the run exits 3."""
    rc, lines, err = run("--gpus", "2", "--rehearse-cpu", "--backend", "gloo", "--steps", "3", "--warmup", "1", BITNUC_BENCH_PLANT_BAD_SLOT="1")
    # the rank exits 3 (as for parity_vs_oracle); the launcher folds a failed rank into its own non-zero code and names the FIRST failure it saw as the root
    # cause -- usually rank 0's "exitcode: 3", now and then its peer's, torn down while rank 0 was exiting (seen once in round 5): the line below is the evidence
    assert rc != 0, (rc, err[-2000:])
    line = json.loads(lines[0])
    assert line["allgather_packed"]["own_slot_ok"] is True and line["allgather_packed"]["all_slots_ok"] is False
    assert line["allgather_packed"]["first_bad_slot"] == 1  # rank 0 prints the line: the wrong word sits in its peer's slot


def test_stalled_collective_is_reported_and_fails():
    rc, lines, err = run("--gpus", "2", "--rehearse-cpu", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                         "--inject-stall", "1", "--dist-timeout", "30")
    assert rc != 0, "a stalled collective must not exit 0"
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["stalled"] is True and "allgather" in line["stalled_stage"]


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--rehearse-cpu"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
