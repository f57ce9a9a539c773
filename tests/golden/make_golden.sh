#!/bin/bash
# Regenerates the golden fixtures in tests/golden/ from the COMPILED REFERENCE
# (oracle/_ref/ref_driver = /root/reference sources built in place by oracle/Makefile,
# driven by oracle/ref_driver.cpp).  Only possible where /root/reference exists.
# The fixtures are data (inputs are re-creatable from oracle/synth.h seeds; outputs are
# what the reference computed); no reference source is stored.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
make -C "$ROOT/oracle" ref >/dev/null
DRV="$ROOT/oracle/_ref/ref_driver"
TMP="$(mktemp -d)"; cd "$TMP"   # the reference writes agent_00_* log files into cwd

# G-small: mixed bounded/unbounded actions, terminated + truncated episodes, every tap
"$DRV" fixture "$HERE/small_mixed.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 \
   lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 ckpt="$TMP/ck_small" pack=4 memck="$TMP/mem_small"
# G-traj: 1200 steps across the 1000-step recompute / Retrace sweep / reward-stat update
"$DRV" fixture "$HERE/traj_1200.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 \
   lenMin=5 lenMax=40 pTerm=0.5 nSteps=1200 tapSteps=2 gradSteps=1000 retSteps=999,1000,1200 \
   maxObs=2000 minObs=500 epsAnneal=5e-7
# G-deep: three hidden layers of unequal width (parametric residual with min() width), Tanh, unbounded
"$DRV" fixture "$HERE/deep_tanh.bin" dimS=9 dimA=3 bounded=000 layers=24,16,8 nnFunc=Tanh batch=8 nEps=20 \
   lenMin=3 lenMax=30 pTerm=0.3 nSteps=6 gradSteps=1,6 maxObs=1000 minObs=100 nnLambda=1e-4 learnrate=1e-3 ckpt="$TMP/ck_deep"
# G-ns: the north-star network/batch shape (17/6, 2x256 SoftSign, B=256) on a 300-episode replay
"$DRV" fixture "$HERE/ns_shape.bin" dimS=17 dimA=6 layers=256,256 batch=256 nEps=300 \
   lenMin=150 lenMax=250 pTerm=0.2 nSteps=4 gradSteps=1 maxObs=100000 minObs=1000
# G-racer: RACER with the Gaussian advantage head (Math/Gaus_advantage.h), same replay as G-small
"$ROOT/oracle/_ref/ref_driver_racer" fixture "$HERE/racer_gauss.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 \
   lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500
# G-racer-traj: the same head across the 1000-step sweep (Retrace with non-zero stored advantages)
"$ROOT/oracle/_ref/ref_driver_racer" fixture "$HERE/racer_traj_1200.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 \
   lenMin=5 lenMax=40 pTerm=0.5 nSteps=1200 tapSteps=2 gradSteps=1000 retSteps=999,1000,1200 \
   maxObs=2000 minObs=500 epsAnneal=5e-7
# G-lstm: RACER (Gaussian advantage) on two LSTM layers, truncated BPTT over 8 steps (the RACER_RNN.json family, BASELINE config 4)
"$ROOT/oracle/_ref/ref_driver_racer" fixture "$HERE/racer_lstm.bin" dimS=5 dimA=2 bounded=10 layers=32,32 nnType=LSTM nnFunc=Tanh bptt=8 \
   batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 ckpt="$TMP/ck_lstm"
# G-mgu: V-RACER on two MGU layers (what a partially observable MDP gets with the default nnType, Approximator.cpp:221-223)
"$DRV" fixture "$HERE/vracer_mgu.bin" dimS=5 dimA=2 bounded=10 layers=32,32 nnType=MGU nnFunc=Tanh bptt=8 \
   batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 ckpt="$TMP/ck_mgu"
# G-discrete: RACER with Discrete_policy / Discrete_advantage (one action variable, 4 options)
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/racer_discrete.bin" dimS=5 dimA=1 nOpt=4 layers=32,32 batch=16 nEps=30 \
   lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 ckpt="$TMP/ck_disc" pack=4 memck="$TMP/mem_disc"
# G-conv-small: discrete RACER behind two convolutional layers (8x8x16 -> k4 -> 5x5x32 -> k3 -> 3x3x64; shapes 6 and 7 of
# Network/Builder.cpp:189-203) on states of 1 + 3 appended observations; (episode, t >= 3) pairs from the harness' sampler;
# "lean": parameter-sized vectors as every 53rd element + sums, the first gradient whole
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/conv_small.bin" dimS=256 dimA=1 nOpt=4 nApp=3 "conv=8,8,16,32,4,1;5,5,32,64,3,1" \
   layers=64 nnFunc=Tanh batch=16 nEps=30 lenMin=6 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 \
   lean=1 ckpt="$TMP/ck_conv"
# G-atari: the RACER_atari.json shape (BASELINE config 5): 84x84 frames x (1 + 3), four SoftSign convolutions, dense 576 -> 512 (Tanh,
# the code default: the settings file names no nnFunc) + parametric residual, 6 options (Pong), batch 128
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/racer_atari.bin" dimS=7056 dimA=1 nOpt=6 nApp=3 \
   "conv=84,84,4,8,8,4;20,20,8,16,6,2;8,8,16,32,4,1;5,5,32,64,3,1" layers=512 nnFunc=Tanh batch=128 nEps=24 lenMin=8 lenMax=18 pTerm=0.5 \
   nSteps=3 gradSteps=1,3 maxObs=262144 minObs=131072 gamma=0.99 explNoise=0.05 lean=1
# G-act-*: the other six activation functions of makeFunction (Network/Layers/Functions.h:643-668), one small fixture each
for F in LRelu Sigm HardSign SoftPlus ExpPlus Exp; do
  "$DRV" fixture "$HERE/act_$F.bin" dimS=5 dimA=2 bounded=10 layers=16,16 nnFunc=$F batch=8 nEps=12 lenMin=5 lenMax=20 pTerm=0.5 \
     nSteps=4 gradSteps=1,4 maxObs=600 minObs=100
done
# G-evict-*: removal rules other than "oldest" (ERoldSeqFilter; getERfilterAlgo, MemoryProcessing.cpp:261-298): more data than
# maxObs before the first step, so the first steps remove episodes by the rule
for F in farpolfrac maxkldiv minerror; do
  "$DRV" fixture "$HERE/evict_$F.bin" dimS=5 dimA=2 bounded=10 layers=16,16 batch=16 nEps=40 lenMin=5 lenMax=40 pTerm=0.5 \
     nSteps=30 gradSteps=1,30 maxObs=500 minObs=200 erFilter=$F
done
# G-sample-*: the prioritised samplers (dataSamplingAlgo PERrank / PERerr / PERseq; ReplayMemory/Sampling.cpp:101-296): the
# std::discrete_distribution over all stored transitions (episodes) is rebuilt before every one of the 30 tapped minibatches
for F in PERrank PERerr PERseq; do
  "$DRV" fixture "$HERE/sample_$F.bin" dimS=5 dimA=2 bounded=10 layers=16,16 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
     nSteps=30 gradSteps=1,30 maxObs=2000 minObs=300 sampling=$F
done
# G-hp-*: hyper-parameters away from the shipped defaults (Retrace lambda < 1, other clip / tolerance / discount / annealing /
# learning rate / exploration / output-weight / weight-decay values; clipImpWeight < 1 with its first-steps quirk)
"$DRV" fixture "$HERE/hp_odd.bin" dimS=7 dimA=3 bounded=101 layers=24,24 nnFunc=Tanh batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.4 \
   nSteps=40 gradSteps=1,40 maxObs=2000 minObs=300 clip=2 penalTol=0.2 gamma=0.9 lambda=0.8 epsAnneal=1e-4 learnrate=3e-4 \
   explNoise=0.2 outWeightsPrefac=0.01 nnLambda=1e-5
"$DRV" fixture "$HERE/hp_lowclip.bin" dimS=5 dimA=2 bounded=01 layers=16,16 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.6 \
   nSteps=40 gradSteps=1,40 maxObs=2000 minObs=300 clip=0.7 penalTol=0.05 gamma=0.99 lambda=0.95 epsAnneal=5e-3
# G-combos: combinations the other fixtures do not hold: discrete head on LSTM layers (BPTT 5), Gaussian advantage on MGU layers
# (BPTT 3, episodes shorter than the window), one hidden layer of 48 Relu units with four mixed-bound actions, two appended
# observations in front of dense layers (no convolution)
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/discrete_lstm.bin" dimS=6 dimA=1 nOpt=5 layers=24,24 nnType=LSTM nnFunc=Tanh bptt=5 \
   batch=12 nEps=30 lenMin=3 lenMax=25 pTerm=0.5 nSteps=20 gradSteps=1,20 maxObs=2000 minObs=200
"$ROOT/oracle/_ref/ref_driver_racer" fixture "$HERE/gauss_mgu.bin" dimS=6 dimA=2 bounded=01 layers=16,16 nnType=MGU nnFunc=Tanh bptt=3 \
   batch=12 nEps=30 lenMin=2 lenMax=12 pTerm=0.5 nSteps=20 gradSteps=1,20 maxObs=2000 minObs=100
"$DRV" fixture "$HERE/one_layer_relu.bin" dimS=9 dimA=4 bounded=0110 layers=48 nnFunc=Relu batch=20 nEps=30 lenMin=4 lenMax=25 pTerm=0.5 \
   nSteps=20 gradSteps=1,20 maxObs=2000 minObs=200
"$DRV" fixture "$HERE/appended_dense.bin" dimS=6 dimA=2 bounded=10 nApp=2 layers=32,32 batch=16 nEps=30 lenMin=4 lenMax=25 pTerm=0.5 \
   nSteps=20 gradSteps=1,20 maxObs=2000 minObs=200
# G-moving: a replay in motion -- one more episode behind every second gradient step (time stamps counted from minTotObsNum
# observations, placeholder errors from the running statistics) with the budget exceeded, so the oldest episodes leave all along
"$DRV" fixture "$HERE/moving_replay.bin" dimS=5 dimA=2 bounded=10 layers=16,16 batch=16 nEps=25 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=60 gradSteps=1,60 maxObs=600 minObs=200 addEvery=2 rewlog=1 rewdir="$TMP"
# ... and 1200 steps of it across the 1000-step sweep (400 arrivals: the replay turns over several times); its statistics line shows
# that totEp / totObs are the counters as of the last step's update, not the instant's
"$DRV" fixture "$HERE/moving_traj_1200.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=1200 tapSteps=2 gradSteps=1000 retSteps=1000,1200 maxObs=1200 minObs=400 epsAnneal=5e-7 addEvery=3
# G-crowded: batch 64 out of 93 stored transitions -- most draws collide, the sort / unique / redraw loop of Sample_uniform runs many
# rounds per minibatch
"$DRV" fixture "$HERE/crowded_sampler.bin" dimS=5 dimA=2 bounded=10 layers=16,16 batch=64 nEps=6 lenMin=12 lenMax=20 pTerm=0.5 \
   nSteps=25 gradSteps=1,25 maxObs=600 minObs=64
# G-resume: one run writes its network and replay-memory checkpoints after 40 steps, a SECOND reference process restarts from them
# (Learner_approximator::restart; initializeLearner is skipped for a restarted learner) and trains 20 more steps with taps
"$DRV" fixture "$HERE/resume_first.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=40 gradSteps=1,40 maxObs=2000 minObs=500 ckpt="$TMP/resume_ck" memck="$TMP/resume_ck"
"$DRV" fixture "$HERE/resume_second.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=0 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=20 gradSteps=1,20 maxObs=2000 minObs=500 resume="$TMP/resume_ck"
# G-threads: the reference run with THREE OpenMP threads: two more generators are seeded from the main one (ExecutionInfo.cpp:392-393:
# the stream of weights and samples is shifted by two draws), the per-thread gradients are summed by reduceThreadsGrad
"$DRV" fixture "$HERE/threads3.bin" dimS=5 dimA=2 bounded=10 layers=16,16 batch=16 nEps=20 lenMin=5 lenMax=30 pTerm=0.5 \
   nSteps=12 gradSteps=1,12 maxObs=1000 minObs=200 threads=3
# G-hist: the importance-weight histogram the reference prints (MemoryProcessing::histogramImportanceWeights), captured from stdout
"$DRV" fixture "$HERE/hist_small.bin" dimS=5 dimA=2 bounded=10 layers=16,16 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=40 tapSteps=2 gradSteps=40 maxObs=2000 minObs=300 muSpread=0.8 hist=1
# official-vs-manual cross check of the harness itself (weights must be bit-identical)
"$DRV" fixture "$TMP/off.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 \
   lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 maxObs=2000 minObs=500 path=official
python3 - "$HERE/small_mixed.bin" "$TMP/off.bin" <<'PY'
import sys
sys.path.insert(0, sys.argv[0] and __import__('os').path.dirname(sys.argv[1]) + '/..')
from golden_io import load_blob
import numpy as np
a, b = load_blob(sys.argv[1]), load_blob(sys.argv[2])
assert np.array_equal(a['Wfinal'], b['Wfinal']), "manual tap path diverged from the official task-queue path"
print("harness check: manual == official (bit-identical weights)")
PY
ls -la "$HERE"/*.bin

# ---- round 3: the rest of the settings surface of this path -----------------------------------------------------------------
# G-ret-*: returnsEstimator other than Retrace (MemoryProcessing::createReturnEstimator, MemoryProcessing.cpp:418-450) across the
# 1000-step sweep; every fixture from here on also carries the <learner>_stats.txt the run wrote (header + the line of step 1000,
# with its dRet column)
"$DRV" fixture "$HERE/ret_gae.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=1200 tapSteps=2 gradSteps=1000 retSteps=1,999,1000,1200 maxObs=2000 minObs=500 epsAnneal=5e-7 retEst=GAE lambda=0.9
"$DRV" fixture "$HERE/ret_explore.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=1200 tapSteps=2 gradSteps=1000 retSteps=1,999,1000,1200 maxObs=2000 minObs=500 epsAnneal=5e-7 retEst=retraceExplore lambda=0.95 addEvery=7
"$DRV" fixture "$HERE/ret_none.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=12 gradSteps=1,12 retSteps=12 maxObs=2000 minObs=500 retEst=none
# G-stats-2100: plain Retrace over two statistics lines (steps 1000 and 2000): the dRet column and its counters' reset
"$DRV" fixture "$HERE/stats_2100.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=2100 tapSteps=1 gradSteps=2000 retSteps=2000 maxObs=2000 minObs=500 epsAnneal=5e-7
# G-outfunc-*: nnOutputFunc (Approximator.cpp:193,228): the output layer's activation and the pre-images of its initial biases
"$DRV" fixture "$HERE/outfunc_tanh.bin" dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 \
   nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 nnOutputFunc=Tanh
"$ROOT/oracle/_ref/ref_driver_racer" fixture "$HERE/outfunc_lrelu_gauss.bin" dimS=5 dimA=2 bounded=10 layers=24,16,8 nnFunc=Tanh batch=16 nEps=30 \
   lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 nnOutputFunc=LRelu
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/outfunc_sigm_discrete.bin" dimS=5 dimA=1 nOpt=4 layers=32,32 batch=16 nEps=30 \
   lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 nnOutputFunc=Sigm
# G-encoder: encoderLayerSizes (Learner_approximator::createEncoder, :149-166): dense layers of the preprocessing network ahead of
# nnLayerSizes, one network; with a zero entry that createEncoder drops
"$DRV" fixture "$HERE/encoder_dense.bin" dimS=5 dimA=2 bounded=10 encoder=24,0 layers=16,16 nnFunc=Tanh batch=16 nEps=30 lenMin=5 lenMax=40 \
   pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 ckpt="$TMP/ck_enc"
# G-rnn: nnType "RNN": dense layers with a recurrent term (Builder.cpp:76-81, Layer_Base.h:64-113), truncated BPTT over 8 steps
"$DRV" fixture "$HERE/vracer_rnn.bin" dimS=5 dimA=2 bounded=10 layers=32,32 nnType=RNN nnFunc=Tanh bptt=8 \
   batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 ckpt="$TMP/ck_rnn"
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/discrete_rnn.bin" dimS=5 dimA=1 nOpt=4 layers=24 nnType=RNN nnFunc=SoftSign bptt=5 \
   batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500
# G-nature: the convolutional stack of the Atari paper (Builder.cpp:189-194: 84x84x4 -> 32 k8 s4 -> 64 k4 s2 -> 64 k3 s1 -> 3136) in front
# of a 512-unit dense layer, discrete RACER with 4 options, batch 32
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/nature_dqn.bin" dimS=7056 dimA=1 nOpt=4 nApp=3 \
   "conv=84,84,4,32,8,4;20,20,32,64,4,2;9,9,64,64,3,1" layers=512 nnFunc=Tanh batch=32 nEps=16 lenMin=8 lenMax=14 pTerm=0.5 \
   nSteps=3 gradSteps=2,3 maxObs=262144 minObs=131072 gamma=0.99 explNoise=0.05 lean=1
# G-a22: the network combinations of Approximator::buildPreprocessing / buildFromSettings (Approximator.cpp:218-271) the other fixtures
# do not hold.  minT = the harness sampler's first step: nAppendedObs + bptt for recurrent nets behind stacked observations, because the
# reference reads the first steps of a BPTT window through the same unsigned subtraction (it segfaults with smaller t)
#   recurrent layers behind appended observations
"$ROOT/oracle/_ref/ref_driver_racer" fixture "$HERE/lstm_appended.bin" dimS=5 dimA=2 bounded=10 nApp=2 minT=6 layers=32,32 nnType=LSTM nnFunc=Tanh bptt=4 \
   batch=16 nEps=30 lenMin=10 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500 lean=1
#   recurrent layers behind a convolutional stack (every step of the window passes through the convolutions)
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/conv_lstm.bin" dimS=256 dimA=1 nOpt=4 nApp=3 minT=7 "conv=8,8,16,32,4,1;5,5,32,64,3,1" \
   layers=32 nnType=LSTM nnFunc=Tanh bptt=4 batch=16 nEps=30 lenMin=10 lenMax=40 pTerm=0.5 nSteps=6 gradSteps=1,2,6 retSteps=6 maxObs=2000 minObs=500 lean=1
#   state variables beside the image: a second input layer behind the conv stack + JoinLayer (Approximator.cpp:249-259, Builder.cpp:26-46),
#   without and with appended observations (there the image is the first 1024 entries of the stacked vector, whatever they are)
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/conv_extra.bin" dimS=1030 dimA=1 nOpt=4 minT=0 "conv=8,8,16,32,4,1;5,5,32,64,3,1" \
   layers=64 nnFunc=Tanh batch=16 nEps=30 lenMin=6 lenMax=40 pTerm=0.5 nSteps=6 gradSteps=1,2,6 retSteps=6 maxObs=2000 minObs=500 lean=1
"$ROOT/oracle/_ref/ref_driver_discrete" fixture "$HERE/conv_extra_appended.bin" dimS=515 dimA=1 nOpt=4 nApp=1 "conv=8,8,16,32,4,1;5,5,32,64,3,1" \
   layers=48 nnFunc=Tanh batch=16 nEps=30 lenMin=6 lenMax=40 pTerm=0.5 nSteps=4 gradSteps=1,2,4 retSteps=4 maxObs=2000 minObs=500 lean=1
#   recurrent layers of more than 64 cells
"$ROOT/oracle/_ref/ref_driver_racer" fixture "$HERE/lstm_wide.bin" dimS=5 dimA=2 bounded=10 layers=96,80 nnType=LSTM nnFunc=Tanh bptt=6 \
   batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 nSteps=6 gradSteps=1,2,6 retSteps=6 maxObs=2000 minObs=500 lean=1
"$DRV" fixture "$HERE/mgu_wide.bin" dimS=6 dimA=2 bounded=01 layers=128 nnType=MGU nnFunc=Tanh bptt=5 \
   batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 nSteps=6 gradSteps=1,2,6 retSteps=6 maxObs=2000 minObs=500 lean=1
#   a partially observable MDP with nnType left at its default: "RNN" encoder layers under "MGU" layers (Approximator.cpp:221-223, 264-270)
"$DRV" fixture "$HERE/pomdp_encoder.bin" dimS=5 dimA=2 bounded=10 pomdp=1 encoder=24 layers=16,16 nnFunc=Tanh bptt=5 \
   batch=16 nEps=30 lenMin=5 lenMax=40 pTerm=0.5 nSteps=12 gradSteps=1,2,12 retSteps=12 maxObs=2000 minObs=500
# G-two-rank: TWO learners (mpiexec -n 2; learners_train_comm = the world; one fixture per rank: <name>.r0 / .r1).  prompt=1 pins every
# poll of the delayed reductions to "complete" (this step's global sums), prompt=2 to "never at the poll" (the previous step's sums:
# the other end of the reference's timing freedom, Utils/DelayedReductor.cpp:34-60; round 6).  The round-5 pair two_rank_traj.bin was
# recorded before its command was written down here; the line below reproduces its trajectory (traj_beta, traj_nfar) value for value,
# its per-step taps were a different selection, so the committed files are kept.  Odd batch and budgets: rounded up to the learners.
MPIEXEC=${MPIEXEC:-/opt/conda/bin/mpiexec}
TWO="dimS=5 dimA=2 bounded=10 layers=32,32 batch=15 nEps=40 lenMin=8 lenMax=30 pTerm=0.5 nSteps=1003 tapSteps=2 gradSteps=1,2,1000,1003 maxObs=601 minObs=99 epsAnneal=5e-7"
# (cd "$TMP" && "$MPIEXEC" -n 2 "$DRV" fixture "$HERE/two_rank_traj.bin" prompt=1 $TWO)
(cd "$TMP" && "$MPIEXEC" -n 2 "$DRV" fixture "$HERE/two_rank_stale.bin" prompt=2 $TWO)
# ... and eight fully tapped steps (two_rank.bin: this line reproduces the round-5 files byte for byte), in both timings
TWO8="dimS=5 dimA=2 bounded=10 layers=32,32 batch=16 nEps=40 lenMin=8 lenMax=30 pTerm=0.5 nSteps=8 gradSteps=1,2,8 retSteps=8 maxObs=4096 minObs=512"
(cd "$TMP" && "$MPIEXEC" -n 2 "$DRV" fixture "$HERE/two_rank.bin" prompt=1 $TWO8)
rm -rf "$TMP"
