# The hand-derived known-answer tests of SURVEY.md 8c / tests/kat_vectors.py driven through SLAMHip.jl -- the reference's own
# names (`predict`, `update`, `add_features`, `associate`, `compute_association`, `predict_observation`) on the device-resident
# state, in Float64 and Float32.  For a maintainer with Julia >= 1.6 and an MI355X:
#
#     julia slam.jl_amd/runtests.jl            (SLAMHIP_LIB=/path/to/libslamhip.so to override the library's place)
#
# UNVERIFIED in the build container (no julia there: SURVEY 8c); the same vectors run through the same C entry points from
# Python (tests/test_gpu_ekf.py) and from plain C (tests/abi_client.c) on every GPU test run, and
# tests/test_abi_cpu.py::test_julia_binding_names_exported_symbols_with_the_right_arity checks every ccall of the module
# against include/slamhip.h.
using Test
using LinearAlgebra

include(joinpath(@__DIR__, "SLAMHip.jl"))
using .SLAMHip

const R = [0.1^2 0.0; 0.0 (pi / 180)^2]

struct Veh                       # the three fields predict reads (src/ekf.jl:14-16)
    measured_speed::Float64
    measured_gamma::Float64
    wheelbase::Float64
end

# KAT-10's coupled 5 x 5 covariance (tests/kat_vectors.py)
const PC = [0.30 0.05 0.02 0.03 -0.02; 0.05 0.20 -0.01 0.01 0.04; 0.02 -0.01 0.01 0.005 -0.003;
            0.03 0.01 0.005 0.5 0.1; -0.02 0.04 -0.003 0.1 0.4]

for T in (Float64, Float32)
    tol = T === Float64 ? 1e-12 : 2e-6
    @testset "SLAMHip KATs $T" begin
        # KAT-1 / KAT-2: predict_observation, compute_association, associate
        st = EKFSlamState{T}([0.0, 0.0, 0.0, 10.0, 0.0], Matrix(1.0I, 5, 5); max_landmarks = 4)
        z, H = predict_observation(st, 1)
        @test z ≈ [10.0, 0.0] atol = tol
        @test H ≈ [-1.0 0.0 0.0 1.0 0.0; 0.0 -0.1 -1.0 0.0 0.1] atol = tol
        nis, nd = compute_association(st, [10.5, 0.02], R, 1)
        @test isapprox(nis, 0.12477014923494524; rtol = 10tol) && isapprox(nd, 0.843006098545911; rtol = 10tol)
        zf, idf, zn = associate(st, reshape([10.5, 0.02, 60.0, 1.0], 2, 2), R, 4.0, 25.0)
        @test idf == reshape([1], 1, 1) && size(zf) == (2, 1) && size(zn) == (2, 1) && zn[1, 1] == 60.0
        for mode in (:sweep, :grid)                          # the two forms of the gating decide alike
            gate_mode!(st, mode)
            _, idm, znm = associate(st, reshape([10.5, 0.02, 60.0, 1.0], 2, 2), R, 4.0, 25.0)
            @test idm == idf && znm == zn && gate_info(st).form == mode
        end
        gate_mode!(st, :auto)

        # KAT-3: predict from x = 0, P = 0
        st = EKFSlamState{T}(zeros(3), zeros(3, 3); max_landmarks = 4)
        st.x, st.cov = predict(st, Veh(8.0, 0.0, 4.0), [0.25 0.0; 0.0 (3pi / 180)^2], 0.025)      # the reference's call pattern
        @test st.x ≈ [0.2, 0.0, 0.0] atol = tol
        P = st.cov
        @test isapprox(P[1, 1], 1.5625e-4; rtol = 1e-5) && isapprox(P[2, 3], 2.74155678e-5; rtol = 1e-5) && P == P'

        # KAT-4: add_features from x = 0, P = 0
        st = EKFSlamState{T}(zeros(3), zeros(3, 3); max_landmarks = 4)
        st.x, st.cov = add_features(st, reshape([10.0, 0.0], 2, 1), R)
        @test st.x ≈ [0.0, 0.0, 0.0, 10.0, 0.0] atol = tol
        P = st.cov
        @test isapprox(P[4, 4], R[1, 1]; rtol = 1e-5) && isapprox(P[5, 5], 100 * R[2, 2]; rtol = 1e-5) && all(iszero, P[1:3, :])

        # KAT-8: update with P = diag(p), one observation (closed form: the two measurement rows decouple)
        p = [0.5, 0.4, 0.02, 1.0, 2.0]
        st = EKFSlamState{T}([0.0, 0.0, 0.0, 10.0, 0.0], Matrix(Diagonal(p)); max_landmarks = 4)
        st.x, st.cov = update(st, reshape([10.5, 0.02], 2, 1), R, [1])
        P = st.cov
        @test isapprox(P[1, 1], 0.5 - 0.25 / 1.51; rtol = 10tol) && isapprox(P[1, 4], 0.5 / 1.51; rtol = 10tol)
        @test isapprox(st.x[4], 10.0 + 0.5 / 1.51; rtol = 10tol)

        # KAT-11: update at phi = pi/6, landmark off the axes (3-4-5), coupled P: the INFORMATION form as the expectation
        x = [1.0, 2.0, pi / 6, 4.0, 6.0]
        Hh = [-3/5 -4/5 0.0 3/5 4/5; 4/25 -3/25 -1.0 -4/25 3/25]
        v = [0.3, -0.015]
        zz = reshape([5.0 + v[1], atan(4.0, 3.0) - pi / 6 + v[2]], 2, 1)
        Pp = inv(inv(PC) + Hh' * inv(R) * Hh)
        xp = x + Pp * Hh' * inv(R) * v
        t11 = T === Float64 ? 1e-9 : 5e-6
        for form in (:cholesky, :joseph)
            st = EKFSlamState{T}(x, PC; max_landmarks = 4)
            ekf_update!(st, zz, R, [1]; form = form)
            @test st.x ≈ xp atol = 6t11
            @test st.cov ≈ Pp atol = t11 / 2
        end

        # KAT-12: predict at phi = pi/3, g = pi/6 (s = 1, c = 0), v dt = 1, coupled P, one landmark
        q2 = (3pi / 180)^2
        st = EKFSlamState{T}([1.0, 2.0, pi / 3, 4.0, 6.0], PC; max_landmarks = 4)
        ekf_predict!(st, 4.0, pi / 6, 2.0, [0.25 0.0; 0.0 q2], 0.25)
        @test st.x ≈ [1.0, 3.0, pi / 3 + 0.25, 4.0, 6.0] atol = 10tol
        P = st.cov
        @test isapprox(P[1, 1], 0.27 + q2; rtol = 10tol) && isapprox(P[2, 2], 0.20 + 0.25 / 16; rtol = 10tol)
        @test isapprox(P[1, 4], 0.025; rtol = 10tol) && isapprox(P[1, 3], 0.01 - q2 * sqrt(3) / 4; rtol = 100tol)
        @test P[4:5, 4:5] == T.(PC[4:5, 4:5])

        # the fused step and the views that do not download the matrix
        st = EKFSlamState{T}([0.0, 0.0, 0.0, 10.0, 0.0], Matrix(Diagonal(p)); max_landmarks = 4)
        a = observe!(st, reshape([10.5, 0.02, 60.0, 1.0], 2, 2), R, 4.0, 25.0)
        @test a == Int32[1, -1] && length(st) == 7
        @test cov_diag(st) ≈ diag(st.cov) && cov_block(st, 4:5, 1:3) == st.cov[4:5, 1:3]
        @test size(landmark_blocks(st)) == (3, 2) && size(feature_ellipses(st)) == (5, 2)
    end
end

@testset "SLAMHip FastSLAM step" begin
    Q = [0.25 0.0; 0.0 (3pi / 180)^2]
    pf = PFSlamState{Float32}(4096, 8; seed = 1234)
    set_pose!(pf, [0.0, 0.0, 0.3])
    lm = hcat([[20cos(0.8l), 20sin(0.8l)] for l in 0:7]...)
    init_landmarks!(pf, lm, 0.01, 0.1)
    ids = Int32[1, 4]
    z = hcat([[hypot(lm[1, i], lm[2, i]), atan(lm[2, i], lm[1, i]) - 0.3] for i in ids]...)
    for _ in 1:3
        step_async!(pf, 1.0, 0.0, 4.0, Q, 0.025, z, ids, R)
    end
    out = flush!(pf)
    @test out[4] == 3 && 1 < out[1] <= 4096
    @test all(isfinite, mean_pose(pf))
end
