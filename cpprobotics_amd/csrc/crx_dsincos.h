// crx_dsincos.h — double-precision sin / cos for the crx engine (host + gfx950 device), bit-identical to glibc 2.35's
// sin() / cos() on FMA-capable x86-64 (the __sin_fma / __cos_fma variants its ifunc selects there) for |x| < 105414336.
//
// The reference's Frenet planner evaluates std::cos(iyaw + M_PI/2.0) and std::sin(iyaw + M_PI/2.0) with a float iyaw
// (/root/reference/src/frenet_optimal_trajectory.cpp:111-112): double-precision libm calls — two SEPARATE calls, cos then sin:
// the reference is built without -O (CMakeLists.txt:4-6), so GCC does not merge them into sincos(), whose glibc
// implementation rounds differently from sin()/cos() in about one call in a thousand.  OCML's device sin/cos are not glibc's.
//
// This header restates glibc's published algorithm (IBM Accurate Mathematical Library, sysdeps/ieee754/dbl-64/s_sin.c with
// sincostab.c / usncs.h / trigo.h as of glibc 2.28 ... 2.35: the slow paths of older versions are gone): for 2^-26 <= |x| < 0.855469
// a Taylor polynomial (|x| < 0.126) or a 1/128-spaced table of sin / cos with short correction polynomials; for
// 0.855469 <= |x| < 2.426265 the same on pi/2 - |x|; for larger arguments a reduction by multiples of pi/2 carried in two
// doubles.  The placement of the fused multiply-adds is the one GCC gave glibc's FMA build, read from the disassembly of the
// libm.so.6 of this image (every constant and the 440-entry table were taken from / cross-checked against its .rodata).
// Verified bit-identical to that libm's sin() and cos() on EVERY argument the planner can form: (double)f + M_PI/2 for all
// 2.16e9 floats f in [-pi, pi] and a margin (tests/tools/dsincos_exhaustive.cpp: 0 mismatches; tests/test_dsincos.py runs a
// strided subset and random doubles of every branch).  On a host whose libm is not the FMA flavour (no FMA / AVX2) the oracle's
// libm calls round differently in rare last-bit cases; every host of this project (and the GPU boxes' EPYC) is FMA-capable.
//
// Notice carried over from the glibc sources this restates (s_sin.c, sincostab.c):
//   IBM Accurate Mathematical Library, written by International Business Machines Corp.
//   Copyright (C) 2001-2022 Free Software Foundation, Inc.
//   This program is free software; you can redistribute it and/or modify it under the terms of the GNU Lesser General Public
//   License as published by the Free Software Foundation; either version 2.1 of the License, or (at your option) any later version.
// (restated from the published algorithm and its table of constants, not a copy of the source text)
#pragma once
#include <stdint.h>
#include "crx_trig.h"   // CRX_HD

namespace crx {

// __sincostab: for k = 0 .. 109, (sin, its low part, cos, its low part) of k/128 — entry 4k .. 4k+3
constexpr int kDsincosTabLen = 440;
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const uint64_t kDsincosTab[kDsincosTabLen] = {
    0x0000000000000000ull, 0x0000000000000000ull, 0x3ff0000000000000ull, 0x0000000000000000ull,
    0x3f7fffeaaaaeeeefull, 0xbc1e45e2ec67b77cull, 0x3fefffc000155552ull, 0x3c8f4a01a0196daeull,
    0x3f8fffaaaaeeeed5ull, 0xbc02ab639a9f0777ull, 0x3fefff000155549full, 0x3c828a28a03a5ef3ull,
    0x3f97ff7001033255ull, 0x3bfefe2b51527336ull, 0x3feffdc006bff7e6ull, 0x3c8ae6dae86977bdull,
    0x3f9ffeaaaeeee86full, 0xbc3cd406fb224ae2ull, 0x3feffc00155527d3ull, 0xbc83b54492d89b5bull,
    0x3fa3feb2b12d45d5ull, 0x3c34ec54203d1c11ull, 0x3feff9c03414a7baull, 0x3c6991f4be6c59bfull,
    0x3fa7fdc01032fba9ull, 0xbc4599bdf46e997aull, 0x3feff7006bfdf99full, 0xbc78b3b560648d5full,
    0x3fabfc6d78586dacull, 0x3c18e4fd03dbf236ull, 0x3feff3c0c8103a31ull, 0x3c74856dbddc0e66ull,
    0x3faffaaaeeed4edbull, 0xbc42d16d32684b69ull, 0x3feff0015549f4d3ull, 0x3c8328387b99426full,
    0x3fb1fc343d808befull, 0xbc5f3d32e6f3be4full, 0x3fefebc222a8ef9full, 0x3c57934934f54c77ull,
    0x3fb3facb12d1755bull, 0xbc5921915299468cull, 0x3fefe7034129ef6full, 0xbc6cbf4337c96f97ull,
    0x3fb5f911fd10b737ull, 0xbc50184f02be9102ull, 0x3fefe1c4c3c873ebull, 0xbc35a9c9057c4a02ull,
    0x3fb7f701032550e4ull, 0x3c3afc2d1800501aull, 0x3fefdc06bf7e6b9bull, 0x3c831902b535f8dbull,
    0x3fb9f4902d55d1f9ull, 0x3c52696d7eac1dc1ull, 0x3fefd5c94b43e000ull, 0xbc62e768cb4f92f9ull,
    0x3fbbf1b78568391dull, 0x3c5e91841dea4cc8ull, 0x3fefcf0c800e99b1ull, 0x3c6ea3d786d186acull,
    0x3fbdee6f16c1cce6ull, 0xbc450f8e2fb71673ull, 0x3fefc7d078d1bc88ull, 0x3c8075d2447db685ull,
    0x3fbfeaaeee86ee36ull, 0xbc4afcb2bcc6f03bull, 0x3fefc015527d5bd3ull, 0x3c8b68f35094efb8ull,
    0x3fc0f3378ddd71d1ull, 0x3c6d8468724f0f9eull, 0x3fefb7db2bfe0695ull, 0x3c821dadf4f65ab1ull,
    0x3fc1f0d3d7afceafull, 0xbc66ef95099769a5ull, 0x3fefaf22263c4bd3ull, 0xbc552ace133a2769ull,
    0x3fc2ee285e4ab88full, 0xbc6e4d0f05dee058ull, 0x3fefa5ea641c36f2ull, 0x3c404da6ed17cc7cull,
    0x3fc3eb312c5d66cbull, 0x3c647d666b66cb91ull, 0x3fef9c340a7cc428ull, 0x3c8c5b6b063b7462ull,
    0x3fc4e7ea4dc5f27bull, 0x3c5949db2ac072fcull, 0x3fef91ff40374d01ull, 0xbc67d03f4d3a9e4cull,
    0x3fc5e44fcfa126f3ull, 0xbc66f443063f89b6ull, 0x3fef874c2e1eecf6ull, 0xbc8c6514e1332b16ull,
    0x3fc6e05dc05a4d4cull, 0xbbd32c5c8b81c940ull, 0x3fef7c1afeffde24ull, 0xbc78f55bc47540b1ull,
    0x3fc7dc102fbaf2b5ull, 0x3c45ab50e23c97c3ull, 0x3fef706bdf9ece1cull, 0xbc8698c80c36dcb4ull,
    0x3fc8d7632efaa944ull, 0xbc620fa262cbb953ull, 0x3fef643efeb82acdull, 0x3c76b00ac1fe28acull,
    0x3fc9d252d0cec312ull, 0x3c59c43d80b1137dull, 0x3fef57948cff6797ull, 0x3c6e3a0d3e03b1d5ull,
    0x3fcaccdb297a0765ull, 0xbc59883b57d6cdebull, 0x3fef4a6cbd1e3a79ull, 0x3c813df0edaebb57ull,
    0x3fcbc6f84edc6199ull, 0x3c69c1a56a7b0cabull, 0x3fef3cc7c3b3d16eull, 0xbc621a3ad28a3494ull,
    0x3fccc0a6588289a3ull, 0xbc6868d09bc87c6bull, 0x3fef2ea5d753ffedull, 0x3c8cc4215f56d583ull,
    0x3fcdb9e15fb5a5d0ull, 0xbc632e20d6cc6fc2ull, 0x3fef20073086649full, 0x3c7b940416c1984bull,
    0x3fceb2a57f8ae5a3ull, 0xbc60be06af572cebull, 0x3fef10ec09c5873bull, 0x3c8d9072762c1283ull,
    0x3fcfaaeed4f31577ull, 0xbc615d88508e32b8ull, 0x3fef01549f7deea1ull, 0x3c8d3c1e99e5cafdull,
    0x3fd0515cbf65155cull, 0xbc79b8c29dfd8ec8ull, 0x3feef141300d2f26ull, 0xbc82aa1b08ded372ull,
    0x3fd0cd00cef36436ull, 0xbc79fb0a0c93e2b5ull, 0x3feee0b1fbc0f11cull, 0xbc4bfd2380bbc3b1ull,
    0x3fd14861aa94ddebull, 0xbc6be881b5b615a4ull, 0x3feecfa744d5efa1ull, 0xbc556d0a4af541d0ull,
    0x3fd1c37d64c6b876ull, 0x3c746076fe0dcff5ull, 0x3feebe214f76efa8ull, 0xbc802f9f12ba543eull,
    0x3fd23e52111aaf36ull, 0xbc74f080334eff18ull, 0x3feeac2061bbaf4full, 0x3c62c1d53e94658dull,
    0x3fd2b8ddc43eb49full, 0x3c61553899f2d807ull, 0x3fee99a4c3a7cd83ull, 0xbc82264b1bc53ce8ull,
    0x3fd3331e94049f87ull, 0x3c7e0cb6b40c302cull, 0x3fee86aebf29a9edull, 0x3c89397afdbb58a7ull,
    0x3fd3ad129769d3d8ull, 0x3c003d5504878398ull, 0x3fee733ea0193d40ull, 0xbc86428b3546ce13ull,
    0x3fd426b7e69ee697ull, 0xbc7f09c75705c59full, 0x3fee5f54b436e9d0ull, 0x3c87eb0fd02fc8bcull,
    0x3fd4a00c9b0f3d20ull, 0x3c7823ba6bb08eadull, 0x3fee4af14b2a449cull, 0xbc868ca02e8a6833ull,
    0x3fd5190ecf68a77aull, 0x3c7b357155eef0f3ull, 0x3fee3614b680d6a5ull, 0xbc727793aa015237ull,
    0x3fd591bc9fa2f597ull, 0x3c67c74bac3fe0cbull, 0x3fee20bf49acd6c1ull, 0xbc5660aec7ef636cull,
    0x3fd60a1429078775ull, 0x3c5b1fd80ba89133ull, 0x3fee0af15a03dbceull, 0x3c5fe8e702771ae6ull,
    0x3fd682138a38d7f7ull, 0xbc7d889202444aadull, 0x3fedf4ab3ebd875eull, 0xbc8e2d8a7e6736c4ull,
    0x3fd6f9b8e33a0255ull, 0x3c742bc14ee9da0dull, 0x3feddded50f228d6ull, 0xbc6e80c8d42ba2bfull,
    0x3fd7710255764214ull, 0xbc66ead7314bb6ceull, 0x3fedc6b7eb995912ull, 0x3c54b364776dcd35ull,
    0x3fd7e7ee03c86d4eull, 0xbc7b63bcdabf5af2ull, 0x3fedaf0b6b888e83ull, 0x3c8a249e2b5e5ceaull,
    0x3fd85e7a12826949ull, 0x3c78a40e9b5face0ull, 0x3fed96e82f71a9dcull, 0x3c8ff61bd5d2039dull,
    0x3fd8d4a4a774992full, 0x3c744a02ea766326ull, 0x3fed7e4e97e17b4aull, 0xbc63b770352bed94ull,
    0x3fd94a6be9f546c5ull, 0xbc769ce13e683f58ull, 0x3fed653f073e4040ull, 0xbc876236434bec37ull,
    0x3fd9bfce02e80510ull, 0x3c709e39a320b0a4ull, 0x3fed4bb9e1c619e0ull, 0x3c8f34bb77858f61ull,
    0x3fda34c91cc50ccaull, 0xbc5a310e3b50cecdull, 0x3fed31bf8d8d7c06ull, 0x3c7e60dd3089cbddull,
    0x3fdaa95b63a09277ull, 0xbc66293eb13c0381ull, 0x3fed1750727d94f0ull, 0x3c80d52b1ec1a48eull,
    0x3fdb1d8305321617ull, 0xbc7ae242cb99f519ull, 0x3fecfc6cfa52ad9full, 0x3c88b5b5508f2a0dull,
    0x3fdb913e30dbac43ull, 0xbc7e38ad2f6c3ff1ull, 0x3fece115909a82e5ull, 0x3c81f139bb31109aull,
    0x3fdc048b17b140a3ull, 0x3c619fe6757e9fa7ull, 0x3fecc54aa2b2972eull, 0x3c64ee162ba83a98ull,
    0x3fdc7767ec7fd19eull, 0xbc5eb14d1a3d5826ull, 0x3feca90c9fc67d0bull, 0xbc646a81485e3462ull,
    0x3fdce9d2e3d4a51full, 0xbc62fc8a12dae298ull, 0x3fec8c5bf8ce1a84ull, 0x3c7ab3d1a1590123ull,
    0x3fdd5bca34047661ull, 0x3c728a44a75fc29cull, 0x3fec6f39208be53bull, 0xbc8741dbfbaadb42ull,
    0x3fddcd4c15329c9aull, 0x3c70d4c6e171fd9aull, 0x3fec51a48b8b175eull, 0xbc61bbb43b9aa880ull,
    0x3fde3e56c1582a69ull, 0xbc50a4821099f88full, 0x3fec339eb01ddd81ull, 0xbc8caaf5ee82c5c0ull,
    0x3fdeaee8744b05f0ull, 0xbc5789b43c9b027dull, 0x3fec1528065b7d50ull, 0xbc8892111312e828ull,
    0x3fdf1eff6bc4f97bull, 0x3c717212f8a7525cull, 0x3febf641081e7536ull, 0x3c8b7bd71628a9a1ull,
    0x3fdf8e99e76abc97ull, 0x3c59d950af2d00a3ull, 0x3febd6ea310294f5ull, 0x3c731bbcc88c109dull,
    0x3fdffdb628d2f57aull, 0x3c6f4a992e905b6aull, 0x3febb723fe630f32ull, 0x3c772bd2452d0a39ull,
    0x3fe0362939c69955ull, 0xbc82d8cd78397b01ull, 0x3feb96eeef58840eull, 0x3c545a3cc78fade0ull,
    0x3fe06d3686946e5bull, 0x3c83f5ae4538ff1bull, 0x3feb764b84b704c2ull, 0xbc8f5848c21b389bull,
    0x3fe0a4021e9e1001ull, 0xbc86f643a13914f6ull, 0x3feb553a410c104eull, 0x3c58ff7947027a16ull,
    0x3fe0da8b26b5672eull, 0xbc8a58def0bee909ull, 0x3feb33bba89c8948ull, 0x3c8ea6a51d1f6ca9ull,
    0x3fe110d0c4b69c3bull, 0x3c8d918998809981ull, 0x3feb11d04162a4c6ull, 0x3c71dd561efbc0c2ull,
    0x3fe146d21f8b7f82ull, 0x3c7bf9535e2739a8ull, 0x3feaef78930bd275ull, 0xbc7f836279746f94ull,
    0x3fe17c8e5f2eedb0ull, 0x3c635e57102e2488ull, 0x3feaccb526f69de5ull, 0x3c88fb6a8dd6b6ccull,
    0x3fe1b204acb02fddull, 0xbc5f190c70cbb5ffull, 0x3feaa98688308913ull, 0xbc0b83d607cd5070ull,
    0x3fe1e7343236574cull, 0x3c722a3fa4f41d5aull, 0x3fea85ed4373e02dull, 0x3c69be06385ec792ull,
    0x3fe21c1c1b0394cfull, 0x3c5e5b324b23aa31ull, 0x3fea61e9e72586afull, 0x3c858330e2fd453full,
    0x3fe250bb93788bbbull, 0x3c7ea3d02457bcceull, 0x3fea3d7d0352bdcfull, 0xbc868dbaeca19669ull,
    0x3fe28511c917a067ull, 0xbc801df1d9a16b70ull, 0x3fea18a729aee445ull, 0x3c395e25736c0358ull,
    0x3fe2b91dea88421eull, 0xbc8fa371db216ab0ull, 0x3fe9f368ed912f85ull, 0xbc81d200c5791606ull,
    0x3fe2ecdf279a3082ull, 0x3c8d3557e0e7e37eull, 0x3fe9cdc2e3f25e5cull, 0x3c83f99112993f62ull,
    0x3fe32054b148bc4full, 0x3c8f6b42095a135bull, 0x3fe9a7b5a36a6514ull, 0x3c8722cfcc9fa7a9ull,
    0x3fe3537db9be0367ull, 0x3c6b327e7af040f0ull, 0x3fe98141c42e1310ull, 0x3c8d1ff80488f08dull,
    0x3fe386597456282bull, 0xbc710fada93b07a8ull, 0x3fe95a67e00cb1fdull, 0xbc80befda21f862dull,
    0x3fe3b8e715a2840aull, 0xbc797653a7d2f07bull, 0x3fe93328926d9e92ull, 0xbc8bb77003600cdaull,
    0x3fe3eb25d36cd53aull, 0xbc5be570e1570fc0ull, 0x3fe90b84784ddaf7ull, 0xbc70feb10ab93b87ull,
    0x3fe41d14e4ba6790ull, 0x3c84608fd287ecf5ull, 0x3fe8e37c303d9ad1ull, 0xbc6463a4b53d4bf8ull,
    0x3fe44eb381cf386bull, 0xbc83ed6c1e6a5505ull, 0x3fe8bb105a5dc900ull, 0x3c8863e03e9474c1ull,
    0x3fe48000e431159full, 0xbc8b194a7463ed10ull, 0x3fe89241985d871full, 0x3c8c48d9c413ed84ull,
    0x3fe4b0fc46aab761ull, 0x3c20da05738cc59aull, 0x3fe869108d77a6c6ull, 0x3c7338ffe2bfe9ddull,
    0x3fe4e1a4e54ed51bull, 0xbc8a492f89b7c76aull, 0x3fe83f7dde701ca0ull, 0xbc4152cf609bc6e8ull,
    0x3fe511f9fd7b351cull, 0xbc85c0e861c48831ull, 0x3fe8158a31916d5dull, 0xbc6de8b90b8228deull,
    0x3fe541facddbb724ull, 0x3c7232c28520d391ull, 0x3fe7eb362eaa1488ull, 0x3c5a1d65a4a5959full,
    0x3fe571a6966d59b3ull, 0x3c5c843b4d0fb198ull, 0x3fe7c0827f09e54full, 0xbc6c73d6d72aee68ull,
    0x3fe5a0fc98813a12ull, 0xbc8d82e2b7d4227bull, 0x3fe7956fcd7f6543ull, 0xbc8ab276e9d45ae4ull,
    0x3fe5cffc16bf8f0dull, 0x3c896cb370eb578aull, 0x3fe769fec655211full, 0xbc6827d5cf8c68c5ull,
    0x3fe5fea4552a9e57ull, 0x3c80b6cef7ee20b7ull, 0x3fe73e30174efba1ull, 0xbc65d3ae3d94ad5full,
    0x3fe62cf49921ac79ull, 0xbc8edd9855b6241aull, 0x3fe712046fa77678ull, 0x3c8425b0a5029c81ull,
    0x3fe65aec2963e755ull, 0x3c8126f96b71053cull, 0x3fe6e57c800cf55eull, 0x3c860286dedbd0a6ull,
    0x3fe6888a4e134b2full, 0xbc86b7d37644d5e6ull, 0x3fe6b898fa9efb5dull, 0x3c715ac786ccf4b2ull,
    0x3fe6b5ce50b7821aull, 0xbc65d5158f702e0full, 0x3fe68b5a92eb6253ull, 0xbc89a91ad985f89cull,
    0x3fe6e2b77c40bde1ull, 0xbc70e729857fad53ull, 0x3fe65dc1fdeb8cbaull, 0xbc597c1b47337c77ull,
    0x3fe70f451d0a8c40ull, 0x3c697ede3885770dull, 0x3fe62fcff20191c7ull, 0x3c6d9143895756efull,
    0x3fe73b7680dea578ull, 0xbc72248306dc12a2ull, 0x3fe6018526f563dfull, 0x3c846ca5e0e432d0ull,
    0x3fe7674af6f7b524ull, 0x3c7e9d3f94ac84a8ull, 0x3fe5d2e255f1f17aull, 0x3c80314104c8892bull,
    0x3fe792c1d0041d52ull, 0xbc8abf05eeb354ebull, 0x3fe5a3e839824077ull, 0x3c8428aa2759be62ull,
    0x3fe7bdda5e28b3c2ull, 0x3c4ad1197ccd0393ull, 0x3fe574978d8e83f2ull, 0x3c8f4714af282d23ull,
    0x3fe7e893f5037959ull, 0x3c80eefbaa650c4cull, 0x3fe544f10f592ca5ull, 0xbc8e7ae8e6c7a62full,
    0x3fe812ede9ae4ba4ull, 0xbc87830adf402ddaull, 0x3fe514f57d7bf3daull, 0x3c747a108073c259ull,
};

struct DsC {
  static constexpr double big = 0x1.8p45;                      // 1.5 * 2^45: u = big + |x| leaves round(|x| * 128) in u's low bits
  static constexpr double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;                     // pi/2 = hp0 + hp1
  static constexpr double mp1 = 0x1.921fb58p+0, mp2 = -0x1.dde973cp-27, pp3 = -0x1.cb3b398p-55, pp4 = -0x1.d747f23e32ed7p-83;
  static constexpr double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p52;
  static constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7;
  static constexpr double cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10;
  static constexpr double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13, s4 = 0x1.71de27b9a7ed9p-19,
                          s5 = -0x1.addffc2fcdf59p-26;
};

CRX_HD uint64_t ds_bits(double x) { union { double d; uint64_t u; } v; v.d = x; return v.u; }
CRX_HD double ds_double(uint64_t u) { union { double d; uint64_t u; } v; v.u = u; return v.d; }
CRX_HD double ds_fabs(double x) { return ds_double(ds_bits(x) & 0x7fffffffffffffffull); }
CRX_HD double ds_copysign(double m, double s) { return ds_double((ds_bits(m) & 0x7fffffffffffffffull) | (ds_bits(s) & 0x8000000000000000ull)); }
CRX_HD double ds_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// tab: the 440 words of kDsincosTab in memory of the caller's choice (a kernel stages them in LDS)
CRX_HD double ds_tab(const uint64_t* tab, int k) { return ds_double(tab[k]); }

// do_cos: cos(x + dx) for |x| < 0.86 from the table entry nearest |x| and correction polynomials in the remainder
CRX_HD double ds_do_cos(double x, double dx, const uint64_t* tab) {
  if (x < 0) dx = -dx;
  const double ax = ds_fabs(x);
  const double u = DsC::big + ax;
  const double xr = (ax - (u - DsC::big)) + dx;
  const double xx = xr * xr;
  const double s = ds_fma(xr * xx, ds_fma(xx, DsC::sn5, DsC::sn3), xr);
  const double c = xx * ds_fma(xx, ds_fma(xx, DsC::cs6, DsC::cs4), DsC::cs2);
  const int k = (int)(uint32_t)ds_bits(u) * 4;
  const double sn = ds_tab(tab, k), ssn = ds_tab(tab, k + 1), cs = ds_tab(tab, k + 2), ccs = ds_tab(tab, k + 3);
  const double cor = ds_fma(-c, cs, ds_fma(-s, ssn, ccs));
  return cs + ds_fma(-s, sn, cor);
}
// TAYLOR_SIN(x*x, x, dx): sin(x + dx) for |x| < 0.126
CRX_HD double ds_taylor_sin(double x, double dx) {
  const double xx = x * x;
  const double poly = ds_fma(xx, ds_fma(xx, ds_fma(xx, ds_fma(xx, DsC::s5, DsC::s4), DsC::s3), DsC::s2), DsC::s1);
  const double t = ds_fma(xx, ds_fma(poly, x, -(0.5 * dx)), dx);
  return x + t;
}
// do_sin: sin(x + dx) for |x| < 0.86
CRX_HD double ds_do_sin(double x, double dx, const uint64_t* tab) {
  if (ds_fabs(x) < 0.126) return ds_taylor_sin(x, dx);
  if (x <= 0) dx = -dx;
  const double ax = ds_fabs(x);
  const double u = DsC::big + ax;
  const double xr = ax - (u - DsC::big);
  const double xx = xr * xr;
  const double s = xr + ds_fma(xr * xx, ds_fma(xx, DsC::sn5, DsC::sn3), dx);
  const double c = ds_fma(xr, dx, xx * ds_fma(xx, ds_fma(xx, DsC::cs6, DsC::cs4), DsC::cs2));
  const int k = (int)(uint32_t)ds_bits(u) * 4;
  const double sn = ds_tab(tab, k), ssn = ds_tab(tab, k + 1), cs = ds_tab(tab, k + 2), ccs = ds_tab(tab, k + 3);
  const double cor = ds_fma(s, cs, ds_fma(-c, sn, ds_fma(s, ccs, ssn)));
  return ds_copysign(sn + cor, x);
}
// reduce_sincos: x = n * pi/2 + (a + da), |a| <= pi/4; returns n mod 4
CRX_HD int ds_reduce(double x, double* a, double* da) {
  const double t = ds_fma(x, DsC::hpinv, DsC::toint);
  const double xn = t - DsC::toint;
  const double y = ds_fma(-xn, DsC::mp2, ds_fma(-xn, DsC::mp1, x));
  const int n = (int)(uint32_t)ds_bits(t) & 3;
  const double t2 = ds_fma(-xn, DsC::pp3, y);
  double db = ds_fma(-DsC::pp3, xn, y - t2);
  const double b = ds_fma(-xn, DsC::pp4, t2);
  db = db + ds_fma(-xn, DsC::pp4, t2 - b);
  *a = b; *da = db;
  return n;
}
CRX_HD double ds_do_sincos(double a, double da, int n, const uint64_t* tab) {
  const double r = (n & 1) ? ds_do_cos(a, da, tab) : ds_do_sin(a, da, tab);
  return (n & 2) ? -r : r;
}

// sin(x), |x| < 105414336 (beyond that glibc switches to another reduction, which is not restated: NaN is returned)
CRX_HD double dsin_(double x, const uint64_t* tab = kDsincosTab) {
  const int k = (int)(uint32_t)(ds_bits(x) >> 32) & 0x7fffffff;
  if (k < 0x3e500000) return x;                                                       // |x| < 2^-26
  if (k < 0x3feb6000) return ds_do_sin(x, 0.0, tab);                                  // |x| < 0.855469
  if (k < 0x400368fd) return ds_copysign(ds_do_cos(DsC::hp0 - ds_fabs(x), DsC::hp1, tab), x);   // |x| < 2.426265
  if (k < 0x419921fb) { double a, da; const int n = ds_reduce(x, &a, &da); return ds_do_sincos(a, da, n, tab); }
  return ds_double(0x7ff8000000000000ull);
}
CRX_HD double dcos_(double x, const uint64_t* tab = kDsincosTab) {
  const int k = (int)(uint32_t)(ds_bits(x) >> 32) & 0x7fffffff;
  if (k < 0x3e400000) return 1.0;                                                     // |x| < 2^-27
  if (k < 0x3feb6000) return ds_do_cos(x, 0.0, tab);
  if (k < 0x400368fd) {
    const double y = DsC::hp0 - ds_fabs(x);
    const double a = y + DsC::hp1;
    const double da = (y - a) + DsC::hp1;
    return ds_do_sin(a, da, tab);
  }
  if (k < 0x419921fb) { double a, da; const int n = ds_reduce(x, &a, &da); return ds_do_sincos(a, da, n + 1, tab); }
  return ds_double(0x7ff8000000000000ull);
}

}  // namespace crx
