# oracle/ekf_ref.jl -- TEST / MEASUREMENT INFRASTRUCTURE, not product code.
#
# A Julia-1.x restatement of the reference's EKF-SLAM hot path, for the `julia` CPU leg of bench.py (BASELINE.md 3 (i)):
# the reference itself is Julia 0.5/0.6 source (`type`, `atan2`, `chol`, `Array{T}(m, n)`) and does not parse on
# Julia >= 1.0, so "the Julia CPU path timed beside the GPU" can only be a restatement like this one.
#
#   UNVERIFIED: the build image has no `julia` binary -- this file has never been executed by its author.  It is
#   written against Julia 1.6+ Base / LinearAlgebra only.  Its arithmetic follows oracle/ekf_ref.py (the NumPy
#   restatement the GPU tests are checked against) function by function; `julia oracle/ekf_ref.jl --selftest` checks
#   the hand-derived known answers KAT-1..KAT-4 and KAT-8 of tests/ (SURVEY.md 8c, tests/kat_vectors.py).
#
# Reference lines restated (relative to the reference's root):
#   wrap_angle            src/common.jl:102-110   (mpi_to_pi: ONE conditional wrap, not a modulo)
#   observation_model     src/common.jl:139-165   (predict_observation: z-hat and the dense 2 x n Jacobian)
#   nis_and_distance      src/data-association.jl:53-63   (compute_association: dense H P H' + R)
#   gated_nearest         src/data-association.jl:1-51    (associate)
#   motion_predict!       src/ekf.jl:8-43         (predict)
#   batch_update          src/ekf.jl:46-77        (update: Cholesky form, P -= W1 W1')
#   augment_state         src/ekf.jl:84-122       (add_features)
# The `*_blocks` twins evaluate the same formulas on the 5 x 5 sub-block the dense products reduce to (what the HIP
# kernels compute); `--bench` times those at the benchmark size, where the dense form is O(nz N n^2) and unusable.
module EKFRef

using LinearAlgebra
using Random
using Printf

export wrap_angle, observation_model, nis_and_distance, gated_nearest, motion_predict!, batch_update, augment_state

"src/common.jl:102-110"
function wrap_angle(a::Float64)
    a > pi && return a - 2pi
    a < -pi && return a + 2pi
    return a
end

"src/common.jl:139-165 -- landmark j (1-based) sits at x[2j+2], x[2j+3]"
function observation_model(x::AbstractVector{Float64}, j::Int)
    f = 3 + 2j - 1
    dx = x[f] - x[1]
    dy = x[f+1] - x[2]
    d2 = dx^2 + dy^2
    d = sqrt(d2)
    zhat = [d, atan(dy, dx) - x[3]]                 # bearing NOT wrapped here (:152)
    H = zeros(2, length(x))
    H[1, 1] = -dx / d;  H[1, 2] = -dy / d;  H[1, 3] = 0.0
    H[2, 1] = dy / d2;  H[2, 2] = -dx / d2; H[2, 3] = -1.0
    H[1, f] = dx / d;   H[1, f+1] = dy / d
    H[2, f] = -dy / d2; H[2, f+1] = dx / d2
    return zhat, H
end

"src/data-association.jl:53-63"
function nis_and_distance(x, P, z::AbstractVector{Float64}, R, j::Int)
    zhat, H = observation_model(x, j)
    v = z .- zhat
    v[2] = wrap_angle(v[2])
    S = H * P * H' + R
    nis = dot(v, inv(S) * v)
    return nis, nis + log(det(S))
end

"src/data-association.jl:1-51 -> (zf 2 x nf, idf 1 x nf, zn 2 x nn)"
function gated_nearest(x, P, z::AbstractMatrix{Float64}, R, gate1::Float64, gate2::Float64)
    nf = (length(x) - 3) ÷ 2
    zf = Matrix{Float64}(undef, 2, 0)
    zn = Matrix{Float64}(undef, 2, 0)
    idf = Matrix{Int}(undef, 1, 0)
    for i in 1:size(z, 2)
        jbest, nbest, outer = 0, Inf, Inf
        for j in 1:nf
            nis, nd = nis_and_distance(x, P, z[:, i], R, j)
            if nis < gate1 && nd < nbest            # (:30-37)
                nbest = nd
                jbest = j
            elseif nis < outer                      # (:38-39)
                outer = nis
            end
        end
        if jbest != 0
            zf = hcat(zf, z[:, i])
            idf = hcat(idf, jbest)
        elseif outer > gate2
            zn = hcat(zn, z[:, i])
        end
    end
    return zf, idf, zn
end

"src/ekf.jl:8-43 (in place, like the reference)"
function motion_predict!(x::Vector{Float64}, P::Matrix{Float64}, v, g, w, Q, dt)
    phi = x[3]
    s, c = sin(g + phi), cos(g + phi)
    vts, vtc = v * dt * s, v * dt * c
    Gv = [1.0 0.0 -vts; 0.0 1.0 vtc; 0.0 0.0 1.0]
    Gu = [dt*c -vts; dt*s vtc; dt*sin(g)/w v*dt*cos(g)/w]
    P[1:3, 1:3] = Gv * P[1:3, 1:3] * Gv' + Gu * Q * Gu'
    if size(P, 1) > 3
        P[1:3, 4:end] = Gv * P[1:3, 4:end]
        P[4:end, 1:3] = P[1:3, 4:end]'
    end
    x[1] += vtc
    x[2] += vts
    x[3] = wrap_angle(phi + v * dt * sin(g) / w)    # the pre-update heading (:39-41)
    return x, P
end

"src/ekf.jl:46-77 -- returns NEW x, P"
function batch_update(x::Vector{Float64}, P::Matrix{Float64}, z::AbstractMatrix{Float64}, R, idf)
    m, n = size(z, 2), length(x)
    m == 0 && return copy(x), copy(P)
    H = zeros(2m, n)
    v = zeros(2m)
    RR = zeros(2m, 2m)
    for i in 1:m
        rows = 2i-1:2i
        zhat, H[rows, :] = observation_model(x, Int(idf[i]))
        v[rows] = [z[1, i] - zhat[1], wrap_angle(z[2, i] - zhat[2])]
        RR[rows, rows] = R
    end
    PHt = P * H'
    S = H * PHt + RR
    S = (S + S') * 0.5
    C = inv(cholesky(Symmetric(S)).U)               # chol(S) of the reference = the upper factor
    W1 = PHt * C
    W = W1 * C'
    return x + W * v, P - W1 * W1'
end

"src/ekf.jl:84-122 -- returns NEW (grown) x, P; later features see earlier ones"
function augment_state(x::Vector{Float64}, P::Matrix{Float64}, z::AbstractMatrix{Float64}, R)
    phi = x[3]
    for i in 1:size(z, 2)
        len = length(x)
        r, b = z[1, i], z[2, i]
        s, c = sin(phi + b), cos(phi + b)
        x = vcat(x, x[1] + r * c, x[2] + r * s)
        Gv = [1.0 0.0 -r*s; 0.0 1.0 r*c]
        Gz = [c -r*s; s r*c]
        Pn = zeros(len + 2, len + 2)
        Pn[1:len, 1:len] = P
        new = len+1:len+2
        Pn[new, new] = Gv * P[1:3, 1:3] * Gv' + Gz * R * Gz'
        Pn[new, 1:3] = Gv * P[1:3, 1:3]
        Pn[1:3, new] = Pn[new, 1:3]'
        if len > 3
            Pn[new, 4:len] = Gv * P[1:3, 4:len]
            Pn[4:len, new] = Pn[new, 4:len]'
        end
        P = Pn
    end
    return x, P
end

# ---- the same formulas on the 5 x 5 sub-block (what the dense products reduce to) -----------------------------------
"nis / nd of every (observation, landmark) pair from the five state rows that matter: 2 x nz x N work, not nz N n^2"
function association_table_blocks(x, P, z, R)
    nf = (length(x) - 3) ÷ 2
    nz = size(z, 2)
    nis = Matrix{Float64}(undef, nz, nf)
    nd = similar(nis)
    Threads.@threads for j in 1:nf
        f = 3 + 2j - 1
        idx = [1, 2, 3, f, f + 1]
        dx = x[f] - x[1]; dy = x[f+1] - x[2]
        d2 = dx^2 + dy^2; d = sqrt(d2)
        H5 = [-dx/d -dy/d 0.0 dx/d dy/d; dy/d2 -dx/d2 -1.0 -dy/d2 dx/d2]
        S = H5 * P[idx, idx] * H5' + R
        Si = inv(S)
        ld = log(det(S))
        b0 = atan(dy, dx) - x[3]
        for i in 1:nz
            v = [z[1, i] - d, wrap_angle(z[2, i] - b0)]
            q = dot(v, Si * v)
            nis[i, j] = q
            nd[i, j] = q + ld
        end
    end
    return nis, nd
end

"the decisions of gated_nearest from the table (lowest index wins a tie: strict <)"
function decisions(nis, nd, gate1, gate2)
    nz, nf = size(nis)
    a = zeros(Int, nz)
    for i in 1:nz
        jbest, nbest, outer = 0, Inf, Inf
        for j in 1:nf
            if nis[i, j] < gate1 && nd[i, j] < nbest
                nbest = nd[i, j]; jbest = j
            elseif nis[i, j] < outer
                outer = nis[i, j]
            end
        end
        a[i] = jbest != 0 ? jbest : (outer > gate2 ? -1 : 0)
    end
    return a
end

"batch_update with P H' assembled from 3 + 2 columns of P per observation; the rank-k down-date is one BLAS syrk-shaped product"
function batch_update_blocks!(x::Vector{Float64}, P::Matrix{Float64}, z, R, idf)
    m, n = size(z, 2), length(x)
    m == 0 && return x, P
    PHt = Matrix{Float64}(undef, n, 2m)
    Hs = Vector{Matrix{Float64}}(undef, m)
    v = zeros(2m)
    for i in 1:m
        j = Int(idf[i]); f = 3 + 2j - 1
        dx = x[f] - x[1]; dy = x[f+1] - x[2]
        d2 = dx^2 + dy^2; d = sqrt(d2)
        H5 = [-dx/d -dy/d 0.0 dx/d dy/d; dy/d2 -dx/d2 -1.0 -dy/d2 dx/d2]
        Hs[i] = H5
        PHt[:, 2i-1:2i] = P[:, [1, 2, 3, f, f + 1]] * H5'
        v[2i-1] = z[1, i] - d
        v[2i] = wrap_angle(z[2, i] - (atan(dy, dx) - x[3]))
    end
    S = zeros(2m, 2m)
    for i in 1:m
        j = Int(idf[i]); f = 3 + 2j - 1
        S[2i-1:2i, :] = Hs[i] * PHt[[1, 2, 3, f, f + 1], :]
        S[2i-1:2i, 2i-1:2i] += R
    end
    S = (S + S') * 0.5
    C = inv(cholesky(Symmetric(S)).U)
    W1 = PHt * C
    x .+= W1 * (C' * v)
    BLAS.syrk!('L', 'N', -1.0, W1, 1.0, P)          # lower triangle; mirrored below
    LinearAlgebra.copytri!(P, 'L')
    return x, P
end

# ---- self-test: the hand-derived known answers of SURVEY.md 8c / tests/kat_vectors.py ------------------------------
function selftest()
    R = [0.01 0.0; 0.0 (pi / 180)^2]
    x = [0.0, 0.0, 0.0, 10.0, 0.0]
    zhat, H = observation_model(x, 1)                                               # KAT-1
    @assert zhat == [10.0, 0.0] && H == [-1.0 0.0 0.0 1.0 0.0; 0.0 -0.1 -1.0 0.0 0.1]
    nis, nd = nis_and_distance(x, Matrix(1.0I, 5, 5), [10.5, 0.02], R, 1)           # KAT-2
    @assert isapprox(nis, 0.12477014923494524; rtol = 1e-12) && isapprox(nd, 0.843006098545911; rtol = 1e-12)
    xp, Pp = motion_predict!(zeros(3), zeros(3, 3), 8.0, 0.0, 4.0, [0.25 0.0; 0.0 (3pi / 180)^2], 0.025)   # KAT-3
    @assert isapprox(xp, [0.2, 0.0, 0.0]; atol = 1e-15) && isapprox(Pp[1, 1], 1.5625e-4; rtol = 1e-12) &&
            isapprox(Pp[2, 3], 2.74155678e-5; rtol = 1e-7)
    xa, Pa = augment_state(zeros(3), zeros(3, 3), reshape([10.0, 0.0], 2, 1), R)    # KAT-4
    @assert isapprox(xa, [0.0, 0.0, 0.0, 10.0, 0.0]; atol = 1e-15) && isapprox(Pa[4, 4], R[1, 1]) && isapprox(Pa[5, 5], 100 * R[2, 2])
    @assert wrap_angle(3.5pi) == 1.5pi                                              # KAT-6
    p = [0.5, 0.4, 0.02, 1.0, 2.0]                                                  # KAT-8
    xu, Pu = batch_update(copy(x), Matrix(Diagonal(p)), reshape([10.5, 0.02], 2, 1), R, [1])
    @assert isapprox(Pu[1, 1], 0.5 - 0.25 / 1.51; rtol = 1e-13) && isapprox(Pu[1, 4], 0.5 / 1.51; rtol = 1e-13) &&
            isapprox(xu[4], 10.0 + 0.5 / 1.51; rtol = 1e-14)
    xb, Pb = batch_update_blocks!(copy(x), Matrix(Diagonal(p)), reshape([10.5, 0.02], 2, 1), R, [1])
    @assert isapprox(xb, xu; rtol = 1e-13) && isapprox(Pb, Pu; rtol = 1e-12)
    # KAT-12 (tests/kat_vectors.py): predict at phi = pi/3, g = pi/6 (s = 1, c = 0), v dt = 1, coupled P, one landmark
    Pc = [0.30 0.05 0.02 0.03 -0.02; 0.05 0.20 -0.01 0.01 0.04; 0.02 -0.01 0.01 0.005 -0.003;
          0.03 0.01 0.005 0.5 0.1; -0.02 0.04 -0.003 0.1 0.4]
    q2 = (3pi / 180)^2
    x12, P12 = motion_predict!([1.0, 2.0, pi / 3, 4.0, 6.0], copy(Pc), 4.0, pi / 6, 2.0, [0.25 0.0; 0.0 q2], 0.25)
    @assert isapprox(x12, [1.0, 3.0, pi / 3 + 0.25, 4.0, 6.0]; atol = 1e-14)
    @assert isapprox(P12[1, 1], 0.27 + q2; rtol = 1e-13) && isapprox(P12[2, 2], 0.20 + 0.25 / 16; rtol = 1e-13) &&
            isapprox(P12[1, 4], 0.025; rtol = 1e-13) && isapprox(P12[1, 3], 0.01 - q2 * sqrt(3) / 4; rtol = 1e-12) && P12[4:5, 4:5] == Pc[4:5, 4:5]
    println("selftest ok")
end

# ---- bench: SURVEY.md 8d's synthetic workload, associate + update per step, wall clock ------------------------------
function bench(N::Int, nz::Int, seconds::Float64)
    rng = MersenneTwister(20240601)
    n = 3 + 2N
    L = 100.0 * sqrt(N / 35)
    lm = L .* rand(rng, 2, N)
    x = vcat([L / 2, L / 2, 0.3], vec(lm .+ 0.1 .* randn(rng, 2, N)))
    A = 0.05 .* randn(rng, n, 16)
    P = A * A' + 0.01I
    R = [0.01 0.0; 0.0 (pi / 180)^2]
    dxy = lm .- x[1:2]
    fwd = findall(i -> dxy[1, i] * cos(0.3) + dxy[2, i] * sin(0.3) > 0, 1:N)
    near = fwd[sortperm([sum(abs2, dxy[:, i]) for i in fwd])[1:nz]]
    steps, matched, t0 = 0, 0, time()
    while time() - t0 < seconds
        z = Matrix{Float64}(undef, 2, nz)
        for (k, j) in enumerate(near)
            z[1, k] = sqrt(sum(abs2, dxy[:, j])) + 0.1 * randn(rng)
            z[2, k] = atan(dxy[2, j], dxy[1, j]) - 0.3 + (pi / 180) * randn(rng)
        end
        nis, nd = association_table_blocks(x, P, z, R)
        a = decisions(nis, nd, 4.0, 25.0)
        sel = findall(>(0), a)
        Pw = copy(P)                                   # the reference's update returns a new matrix (:74-76)
        batch_update_blocks!(copy(x), Pw, z[:, sel], R, a[sel])
        steps += 1
        matched += length(sel)
    end
    dt = time() - t0
    @printf("{\"kind\": \"julia\", \"value\": %.3f, \"unit\": \"obs-updates/s\", \"steps_per_s\": %.4f, \"cores\": %d, \"blas_threads\": %d, \"sample\": \"N=%d, %d obs/step, fp64, %d steps in %.1f s (5 x 5 sub-block association + syrk down-date)\"}\n",
            matched / dt, steps / dt, Threads.nthreads(), BLAS.get_num_threads(), N, nz, steps, dt)
end

end # module

if abspath(PROGRAM_FILE) == @__FILE__
    if "--selftest" in ARGS
        EKFRef.selftest()
    elseif length(ARGS) >= 4 && ARGS[1] == "--bench"
        EKFRef.selftest()
        EKFRef.bench(parse(Int, ARGS[2]), parse(Int, ARGS[3]), parse(Float64, ARGS[4]))
    else
        println("usage: julia oracle/ekf_ref.jl --selftest | --bench N nz seconds")
    end
end
